#!/usr/bin/env python3
"""Interleaved A/B sweep of the tile configurations compiled into libpea_hip.so (one process, HIP events).
Usage on the GPU box:  python profiles/sweep_tiles.py [B]     (default B=8, CVPPP shape, K=10)"""
import ctypes
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

pkg = ge.load_package()
synth = importlib.import_module(ge.PKG_NAME + ".utils.synth")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
D, H, W = 16, 544, 544
dev = torch.device("cuda:0")
offsets = pkg.multi_offset([1, 3, 5, 9, 27], 4)
K = len(offsets)
e, t, w, m = synth.synth_inputs_2d(B, D, H, W, offsets, 555)
E, T, Wt, M = (torch.from_numpy(x).to(dev) for x in (e, t, w, m))
op, L = pkg.affinity_op, pkg._lib.lib()
spec = op.AffinitySpec(2, offsets, None, pkg._lib.BORDER_CIRCULAR, pkg._lib.NORM_BX)
desc = op.make_desc(spec, E)
affs = torch.empty(B, K, H, W, device=dev)
lossv = torch.empty(1 + K, device=dev)
G = torch.empty(B, K, H, W, device=dev)
wsb = L.pea_workspace_bytes(ctypes.byref(desc))
work = torch.empty(max(wsb, 4) // 4, device=dev)
dE = torch.empty_like(E)
one = torch.ones((), device=dev)
P = lambda x: ctypes.c_void_p(x.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
fwd = lambda: L.pea_affinity_fwd(ctypes.byref(desc), P(E), None, P(T), P(Wt), P(M), P(affs), P(G), P(lossv), P(work), wsb, st)
bwd = lambda: L.pea_affinity_bwd(ctypes.byref(desc), P(E), None, P(G), P(one), P(dE), None, st)
inf = lambda: L.pea_affinity_infer(ctypes.byref(desc), P(E), None, P(affs), st)


def timed(fn, iters=30):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        rc = fn()
        assert rc == 0, rc
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / iters * 1e3  # us


def setenv(**kw):
    for k in ("PEA_FORCE_DIRECT", "PEA_FWD_CFG", "PEA_BWD_CFG", "PEA_NEAR_R", "PEA_LDS_MAX"):
        os.environ.pop(k, None)
    for k, v in kw.items():
        os.environ[k] = str(v)


setenv(PEA_FORCE_DIRECT=1)
fwd(); bwd(); torch.cuda.synchronize()
ref_affs, ref_loss, ref_dE = affs.clone(), lossv.clone(), dE.clone()
variants = [("direct", dict(PEA_FORCE_DIRECT=1))]
for ci in range(3):
    for r in (9,):
        variants.append(("fwd cfg%d R%d" % (ci, r), dict(PEA_FWD_CFG=ci, PEA_BWD_CFG=99, PEA_NEAR_R=r)))
for ci in range(2):
    for r in (9,):
        variants.append(("bwd cfg%d R%d" % (ci, r), dict(PEA_BWD_CFG=ci, PEA_FWD_CFG=99, PEA_NEAR_R=r)))
res = {}
for rnd in range(3):
    for name, env in variants:
        setenv(**env)
        which = [("fwd", fwd), ("inf", inf)] if name.startswith("fwd") else [("bwd", bwd)] if name.startswith("bwd") else [("fwd", fwd), ("inf", inf), ("bwd", bwd)]
        for kn, fn in which:
            if rnd == 0:
                affs.zero_(); dE.zero_(); lossv.zero_()
                fn(); torch.cuda.synchronize()
                if kn in ("fwd", "inf"):
                    err = (affs - ref_affs).abs().max().item()
                    if kn == "fwd":
                        err = max(err, ((lossv - ref_loss).abs() / ref_loss.abs()).max().item())
                else:
                    err = ((dE - ref_dE).abs().max() / ref_dE.abs().max()).item()
                res.setdefault((name, kn), {"err": err, "t": []})
                timed(fn, 5)
            res[(name, kn)]["t"].append(timed(fn))
px = B * H * W
for (name, kn), r in res.items():
    tmin, tmed = min(r["t"]), sorted(r["t"])[len(r["t"]) // 2]
    ab = {"fwd": 4 * D + 13 * K, "inf": 4 * D + 4 * K, "bwd": 8 * D + 9 * K}[kn]
    print("%-16s %-4s min %8.1f us  med %8.1f us  %6.0f GB/s alg  err %.2e" % (name, kn, tmin, tmed, ab * px / tmin / 1e3, r["err"]))
