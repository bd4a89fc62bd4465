#!/usr/bin/env python3
"""A/B harness for experimental kernels built as csrc/exp/libcross_exp.so: checks the experimental backward against
pea_affinity_bwd of the product library on the same inputs and times both with HIP events.
Usage: python profiles/exp_cross.py [B] [cfg ...]"""
import ctypes
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge

pkg = ge.load_package()
synth = importlib.import_module(ge.PKG_NAME + ".utils.synth")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfgs = [int(a) for a in sys.argv[2:]] or [0, 1]
D, H, W = 16, 544, 544
dev = torch.device("cuda:0")
offsets = pkg.multi_offset([1, 3, 5, 9, 27], 4)
K = len(offsets)
e, t, w, m = synth.synth_inputs_2d(B, D, H, W, offsets, 555)
E, T, Wt, M = (torch.from_numpy(x).to(dev) for x in (e, t, w, m))
op, L = pkg.affinity_op, pkg._lib.lib()
X = ctypes.CDLL(os.path.join(ROOT, "pixel-embedded-affinity_amd", "csrc", "exp", "libcross_exp.so"))
desc = op.make_desc(op.AffinitySpec(2, offsets, None, pkg._lib.BORDER_CIRCULAR, pkg._lib.NORM_BX), E)
affs = torch.empty(B, K, H, W, device=dev)
G = torch.empty(B, K, H, W, device=dev)
lossv = torch.empty(1 + K, device=dev)
wsb = L.pea_workspace_bytes(ctypes.byref(desc))
work = torch.empty(max(wsb, 4) // 4, device=dev)
dE = torch.empty_like(E)
dE2 = torch.empty_like(E)
INV = torch.empty(B, H, W, device=dev)
one = torch.full((), 0.75, device=dev)
P = lambda x: ctypes.c_void_p(x.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
assert L.pea_affinity_fwd(ctypes.byref(desc), P(E), None, P(T), P(Wt), P(M), P(affs), P(G), P(lossv), P(work), wsb, st) == 0
assert L.pea_affinity_bwd(ctypes.byref(desc), P(E), None, P(G), P(one), P(dE), None, st) == 0
assert X.pea_x_inv(ctypes.byref(desc), P(E), P(INV), st) == 0
torch.cuda.synchronize()


def timeit(fn, n=30, rounds=5):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b) / n * 1e3)
    return min(ts), sorted(ts)[len(ts) // 2]


print("old bwd       min %.1f us med %.1f us" % timeit(lambda: L.pea_affinity_bwd(ctypes.byref(desc), P(E), None, P(G), P(one), P(dE), None, st)))
print("inv plane     min %.1f us med %.1f us" % timeit(lambda: X.pea_x_inv(ctypes.byref(desc), P(E), P(INV), st)))
ref = dE.clone()
for cfg in cfgs:
    dE2.zero_()
    rc = X.pea_x_bwd(ctypes.byref(desc), P(E), P(INV), P(G), P(one), P(dE2), cfg, st)
    torch.cuda.synchronize()
    if rc != 0:
        print("cfg %d: rc %d" % (cfg, rc))
        continue
    err = float((dE2 - ref).abs().max() / ref.abs().max())
    print("cfg %d: rel-to-max err vs old bwd %.3e" % (cfg, err))
    print("cfg %d new bwd min %.1f us med %.1f us" % ((cfg,) + timeit(lambda: X.pea_x_bwd(ctypes.byref(desc), P(E), P(INV), P(G), P(one), P(dE2), cfg, st))))
