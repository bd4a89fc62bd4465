#!/usr/bin/env python3
"""where the 3D step's time goes: norm5 (K=12) against its in-plane part (K=8) and its z part (K=4) on 16 x 24 x 1024^2,
and the in-plane part as 24 independent 2D images (circular border) -- fwd_ex / bwd_ex with the 1/norm plane (cross kernels)"""
import ctypes, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
aff = importlib.import_module(ge.PKG_NAME + ".utils.affinity_ours")
dev = torch.device("cuda:0")
op, L = pkg.affinity_op, pkg._lib.lib()
P = lambda x: ctypes.c_void_p(x.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
n5 = aff.axis_offsets_3d(aff.NORM5_SHIFTS)
inpl = [o for o in n5 if o[0] == 0]
zonly = [o for o in n5 if o[0] != 0]
Z, Y, X = 24, 1024, 1024
cases = [("norm5 K=12", 3, 1, n5), ("in-plane K=8", 3, 1, inpl), ("z only K=4", 3, 1, zonly),
         ("2D 24 images K=8", 2, 24, [[o[1], o[2]] for o in inpl])]
if os.environ.get("SMALL_PLANES"):   # the same voxel count as 16 volumes of 24 x 256 x 256: planes 256 KB apart instead of 4 MB
    Y, X = 256, 256
    cases = [("norm5 K=12 256^2 x16", 3, 16, n5), ("in-plane 256^2 x16", 3, 16, inpl), ("z only 256^2 x16", 3, 16, zonly)]
if os.environ.get("ODD_PLANES"):     # plane stride not a power of two
    Y, X = 1040, 1008
    cases = [("norm5 K=12 1040x1008", 3, 1, n5), ("in-plane 1040x1008", 3, 1, inpl), ("z only 1040x1008", 3, 1, zonly)]
for name, nd, B, offs in cases:
    K = len(offs)
    g = torch.Generator(device=dev); g.manual_seed(1)
    shp = (B, 16, Z, Y, X) if nd == 3 else (B, 16, Y, X)
    kshp = (B, K) + shp[2:]
    E = torch.randn(*shp, device=dev, generator=g)
    T = (torch.rand(*kshp, device=dev, generator=g) < 0.7).float()
    Wt = torch.rand(*kshp, device=dev, generator=g) + 0.5
    if nd == 3:
        spec = op.AffinitySpec(3, offs, None, pkg._lib.BORDER_CROP_ZERO, pkg._lib.NORM_CROPPED)
    else:
        spec = op.AffinitySpec(2, offs, None, pkg._lib.BORDER_CIRCULAR, pkg._lib.NORM_BX)
    desc = op.make_desc(spec, E)
    affs = torch.empty(*kshp, device=dev); G = torch.empty_like(affs); lossv = torch.empty(1 + K, device=dev)
    INV = torch.empty((B, 1) + shp[2:], device=dev)
    wsb = L.pea_workspace_bytes(ctypes.byref(desc)); work = torch.empty(max(wsb, 4) // 4, device=dev)
    assert L.pea_workspace_init(ctypes.c_void_p(work.data_ptr()), wsb, None) == 0  # the loss-state block: prepared once
    dE = torch.empty_like(E); one = torch.ones((), device=dev)
    fns = {"fwd": lambda: L.pea_affinity_fwd_ex(ctypes.byref(desc), P(E), None, P(T), P(Wt), None, P(affs), P(G), P(INV), P(lossv), P(work), wsb, st),
           "bwd": lambda: L.pea_affinity_bwd_ex2(ctypes.byref(desc), P(E), None, P(G), P(INV), P(affs), P(one), P(dE), None, st)}
    out = ["cross fwd/bwd %d/%d" % (L.pea_cross_supported(ctypes.byref(desc), 0), L.pea_cross_supported(ctypes.byref(desc), 1))]
    for kn, fn in fns.items():
        for _ in range(3): assert fn() == 0
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(6): assert fn() == 0
        b.record(); b.synchronize()
        out.append("%s %8.1f us" % (kn, a.elapsed_time(b) / 6 * 1e3))
    print("%-20s %s" % (name, "  ".join(out)), flush=True)
    del E, T, Wt, affs, G, dE, INV
    torch.cuda.empty_cache()
