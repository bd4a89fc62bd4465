#!/usr/bin/env python3
"""The full-resolution forward pair of the CVPPP training loop (self loss + detached-EMA cross loss on the same target / weight / mask,
B x 16 x 544^2, K = 10): two pea_affinity_fwd_ex launches against ONE pea_affinity_fwd_dual_ex launch (csrc/pea_xdma_dual.h), ring of
two (two workgroups per CU) and of three (one).  HIP events around 50 calls each, three rounds, variants alternating.
    python profiles/exp_r5_dual.py"""
import os, sys, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
import importlib
pkg = ge.load_package()
synth = importlib.import_module(ge.PKG_NAME + ".utils.synth")
dev = torch.device("cuda:0")
B, D, H, W = int(os.environ.get("EXP_B", "8")), 16, 544, 544
offsets = pkg.multi_offset([1, 3, 5, 9, 27], 4)
K = len(offsets)
e, t, w, m = synth.synth_inputs_2d(B, D, H, W, offsets, seed=555)
E, T, Wt, M = (torch.from_numpy(x).to(dev) for x in (e, t, w, m))
EO = torch.from_numpy(synth.synth_embedding((B, D, H, W), 700)).to(dev)
op, L = pkg.affinity_op, pkg._lib.lib()
d0 = op.make_desc(op.AffinitySpec(2, offsets, None, pkg._lib.BORDER_CIRCULAR, pkg._lib.NORM_BX), E)
dx = op.make_desc(op.AffinitySpec(2, offsets, [2.0, 2.0] + [1.0] * (K - 2), pkg._lib.BORDER_CIRCULAR, pkg._lib.NORM_BX), E)
P = lambda x: None if x is None else ctypes.c_void_p(x.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
wsb = L.pea_workspace_bytes(ctypes.byref(d0))
work = torch.empty(2, wsb // 4, device=dev)
assert L.pea_workspace_init(P(work), 2 * wsb, None) == 0
affs, g0, gx = (torch.empty(B, K, H, W, device=dev) for _ in range(3))
inv = torch.empty(2, B, H, W, device=dev)
inv0 = torch.empty(B, H, W, device=dev)
l0, lx = torch.empty(1 + K, device=dev), torch.empty(1 + K, device=dev)
one, dE = torch.ones((), device=dev), torch.empty_like(E)


def two():
    assert L.pea_affinity_fwd_ex(ctypes.byref(d0), P(E), None, P(T), P(Wt), P(M), P(affs), P(g0), P(inv0), P(l0), P(work[0]), wsb, st) == 0
    assert L.pea_affinity_fwd_ex(ctypes.byref(dx), P(E), P(EO), P(T), P(Wt), P(M), None, P(gx), P(inv), P(lx), P(work[1]), wsb, st) == 0


def dual():
    assert L.pea_affinity_fwd_dual_ex(ctypes.byref(d0), ctypes.byref(dx), P(E), P(EO), P(T), P(Wt), P(M), P(affs), P(g0), P(gx), P(inv[0]), P(inv[1]),
                                      P(l0), P(lx), P(work[0]), P(work[1]), wsb, st) == 0


def bwd():
    assert L.pea_affinity_bwd_dual_ex(ctypes.byref(d0), P(E), P(EO), P(g0), P(gx), P(inv[0]), P(inv[1]), P(one), P(one), P(dE), st) == 0


BWD_US = [0.0]


def timed(fn, n=50):
    for _ in range(5):
        fn(); bwd()
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(n)]
    for a, b, c in ev:
        a.record(); fn(); b.record(); bwd(); c.record()
    torch.cuda.synchronize()
    BWD_US[0] = sum(b.elapsed_time(c) for a, b, c in ev) / n * 1e3
    return sum(a.elapsed_time(b) for a, b, c in ev) / n * 1e3


two(); ref = [x.clone() for x in (affs, g0, gx, inv, l0, lx)]
for rnd in range(3):
    for name, env, fn in (("two launches", None, two), ("dual ring 2", "2", dual), ("dual ring 3", "3", dual), ("dual halves", "4", dual)):
        if env is not None:
            os.environ["PEA_FWD_DUAL"] = env
            pkg._lib.reload_env()
        us = timed(fn)
        same = all(torch.equal(a, b) for a, b in zip((affs, g0, gx, inv, l0, lx), ref))
        print("round %d  %-14s %7.1f us (forward pair inside the alternating forward / dual backward step)   bit-equal to the two launches: %s   [pair backward %.1f us]" % (rnd, name, us, same, BWD_US[0]), flush=True)
