#!/usr/bin/env python3
"""Run one entry point (fwd | bwd | inf) a few times at the CVPPP bench shape, for rocprofv3 --pmc passes.
Usage: python profiles/one_kernel.py <fwd|bwd|inf> [iters] [B]   (tile config via PEA_* env vars)"""
import ctypes
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

pkg = ge.load_package()
synth = importlib.import_module(ge.PKG_NAME + ".utils.synth")
which = sys.argv[1]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
D, H, W = 16, 544, 544
dev = torch.device("cuda:0")
offsets = pkg.multi_offset([1, 3, 5, 9, 27], 4)
K = len(offsets)
e, t, w, m = synth.synth_inputs_2d(B, D, H, W, offsets, 555)
E, T, Wt, M = (torch.from_numpy(x).to(dev) for x in (e, t, w, m))
op, L = pkg.affinity_op, pkg._lib.lib()
desc = op.make_desc(op.AffinitySpec(2, offsets, None, pkg._lib.BORDER_CIRCULAR, pkg._lib.NORM_BX), E)
affs = torch.empty(B, K, H, W, device=dev)
G = torch.empty(B, K, H, W, device=dev)
lossv = torch.empty(1 + K, device=dev)
wsb = L.pea_workspace_bytes(ctypes.byref(desc))
work = torch.empty(max(wsb, 4) // 4, device=dev)
assert L.pea_workspace_init(ctypes.c_void_p(work.data_ptr()), wsb, None) == 0  # the loss-state block: prepared once
dE = torch.empty_like(E)
one = torch.ones((), device=dev)
P = lambda x: ctypes.c_void_p(x.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
LAB = torch.from_numpy(synth.synth_labels(B, (1, H, W), 555)[:, 0].copy()).to(dev)
WTAB = torch.empty(B * K * 2, device=dev)
CNTB = L.pea_targets_workspace_bytes(ctypes.byref(desc))
CNT = torch.empty(CNTB // 4, dtype=torch.int32, device=dev)
assert L.pea_label_weights(ctypes.byref(desc), P(LAB), 5, P(WTAB), P(CNT), CNTB, st) == 0
fns = {
    "labels_noaffs": lambda: L.pea_affinity_fwd_bwd_labels(ctypes.byref(desc), P(E), None, P(LAB), P(WTAB), 5, None, P(lossv), None, P(dE), P(work), wsb, st),
    "fwd_nog": lambda: L.pea_affinity_fwd(ctypes.byref(desc), P(E), None, P(T), P(Wt), P(M), P(affs), None, P(lossv), P(work), wsb, st),
    "gentgt": lambda: L.pea_gen_targets(ctypes.byref(desc), P(LAB), 1, P(T), P(M), P(Wt), P(CNT), CNTB, st),
    "labw": lambda: L.pea_label_weights(ctypes.byref(desc), P(LAB), 5, P(WTAB), P(CNT), CNTB, st),
    "labels": lambda: L.pea_affinity_fwd_bwd_labels(ctypes.byref(desc), P(E), None, P(LAB), P(WTAB), 5, P(affs), P(lossv), None, P(dE), P(work), wsb, st),
    "fwd": lambda: L.pea_affinity_fwd(ctypes.byref(desc), P(E), None, P(T), P(Wt), P(M), P(affs), P(G), P(lossv), P(work), wsb, st),
    "bwd": lambda: L.pea_affinity_bwd(ctypes.byref(desc), P(E), None, P(G), P(one), P(dE), None, st),
    "inf": lambda: L.pea_affinity_infer(ctypes.byref(desc), P(E), None, P(affs), st),
}
INV = torch.empty(B, H, W, device=dev)
LSB = L.pea_labels_scratch_bytes(ctypes.byref(desc))
LSCR = torch.empty(max(LSB, 4) // 4, device=dev)
# the two-launch labels step (labels-in forward on the cross kernels + cross backward) and the forward / backward pair with the plane
fns["labels2"] = lambda: L.pea_affinity_fwd_bwd_labels_ex(ctypes.byref(desc), P(E), None, P(LAB), P(WTAB), 5, P(affs), P(lossv), None, P(dE),
                                                           P(work), wsb, P(LSCR), LSB, st)
fns["fwd_ex"] = lambda: L.pea_affinity_fwd_ex(ctypes.byref(desc), P(E), None, P(T), P(Wt), P(M), P(affs), P(G), P(INV), P(lossv), P(work), wsb, st)
if which == "bwd":
    fns["fwd"]()
for _ in range(iters):
    assert fns[which]() == 0
torch.cuda.synchronize()
