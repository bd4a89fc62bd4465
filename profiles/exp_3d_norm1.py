#!/usr/bin/env python3
"""norm1 (K = 3: one step along z, y, x) on a 24 x 1024^2 sub-volume: the 3D cross kernels (z by global gathers) against the unit-box
kernels (PEA_FWD_XDMA=0 PEA_BWD_XDMA=0 makes the dispatcher skip the cross kernels).  One setting per process."""
import ctypes, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
dev = torch.device("cuda:0")
op, L = pkg.affinity_op, pkg._lib.lib()
P = lambda x: ctypes.c_void_p(x.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
offs = [[-1, 0, 0], [0, -1, 0], [0, 0, -1]]
Z, Y, X, K = 24, 1024, 1024, 3
g = torch.Generator(device=dev); g.manual_seed(1)
E = torch.randn(1, 16, Z, Y, X, device=dev, generator=g)
T = (torch.rand(1, K, Z, Y, X, device=dev, generator=g) < 0.7).float()
Wt = torch.rand(1, K, Z, Y, X, device=dev, generator=g) + 0.5
spec = op.AffinitySpec(3, offs, None, pkg._lib.BORDER_CROP_ZERO, pkg._lib.NORM_CROPPED)
desc = op.make_desc(spec, E)
affs = torch.empty_like(T); G = torch.empty_like(T); lossv = torch.empty(1 + K, device=dev)
INV = torch.empty(1, 1, Z, Y, X, device=dev)
wsb = L.pea_workspace_bytes(ctypes.byref(desc)); work = torch.empty(max(wsb, 4) // 4, device=dev)
assert L.pea_workspace_init(ctypes.c_void_p(work.data_ptr()), wsb, None) == 0
dE = torch.empty_like(E); one = torch.ones((), device=dev)
fns = {"fwd": lambda: L.pea_affinity_fwd_ex(ctypes.byref(desc), P(E), None, P(T), P(Wt), None, P(affs), P(G), P(INV), P(lossv), P(work), wsb, st),
       "bwd": lambda: L.pea_affinity_bwd_ex(ctypes.byref(desc), P(E), None, P(G), P(INV), P(one), P(dE), None, st),
       "inf": lambda: L.pea_affinity_infer(ctypes.byref(desc), P(E), None, P(affs), st)}
out = []
for kn, fn in fns.items():
    for _ in range(3): assert fn() == 0
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(6): assert fn() == 0
    b.record(); b.synchronize()
    out.append("%s %8.1f us" % (kn, a.elapsed_time(b) / 6 * 1e3))
print("norm1 xdma fwd/bwd = %s/%s  %s  loss %.6f |dE| %.6e" % (os.environ.get("PEA_FWD_XDMA", "1"), os.environ.get("PEA_BWD_XDMA", "1"), "  ".join(out),
                                                                float(lossv[0]), float(dE.abs().sum())), flush=True)
