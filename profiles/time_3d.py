#!/usr/bin/env python3
"""time fwd / bwd / inf on 3D sub-volumes (CROP_ZERO border, cropped normaliser): the norm5 stencil at the reference
training shape and at an AC3/AC4-sized sub-volume, and the 26-neighbourhood of BASELINE configs[3]"""
import ctypes, importlib, itertools, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
synth = importlib.import_module(ge.PKG_NAME + ".utils.synth")
aff = importlib.import_module(ge.PKG_NAME + ".utils.affinity_ours")
dev = torch.device("cuda:0")
op, L = pkg.affinity_op, pkg._lib.lib()
P = lambda x: ctypes.c_void_p(x.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
n26 = [o for o in itertools.product((-1, 0, 1), repeat=3) if o != (0, 0, 0)]
cases = [("norm5 K=12, B=2 x 16 x 18x160x160", 2, 18, 160, 160, aff.axis_offsets_3d(aff.NORM5_SHIFTS)),
         ("norm5 K=12, B=1 x 16 x 24x1024x1024", 1, 24, 1024, 1024, aff.axis_offsets_3d(aff.NORM5_SHIFTS)),
         ("26-neighbourhood, B=1 x 16 x 24x1024x1024", 1, 24, 1024, 1024, n26)]
for name, B, Z, Y, X, offs in cases:
    K = len(offs)
    g = torch.Generator(device=dev); g.manual_seed(1)
    E = torch.randn(B, 16, Z, Y, X, device=dev, generator=g)
    T = (torch.rand(B, K, Z, Y, X, device=dev, generator=g) < 0.7).float()
    Wt = torch.rand(B, K, Z, Y, X, device=dev, generator=g) + 0.5
    desc = op.make_desc(op.AffinitySpec(3, offs, None, pkg._lib.BORDER_CROP_ZERO, pkg._lib.NORM_CROPPED), E)
    affs = torch.empty(B, K, Z, Y, X, device=dev); G = torch.empty_like(affs); lossv = torch.empty(1 + K, device=dev)
    wsb = L.pea_workspace_bytes(ctypes.byref(desc)); work = torch.empty(max(wsb, 4) // 4, device=dev)
    assert L.pea_workspace_init(ctypes.c_void_p(work.data_ptr()), wsb, None) == 0  # the loss-state block: prepared once
    dE = torch.empty_like(E); one = torch.ones((), device=dev)
    fns = {"fwd": lambda: L.pea_affinity_fwd(ctypes.byref(desc), P(E), None, P(T), P(Wt), None, P(affs), P(G), P(lossv), P(work), wsb, st),
           "bwd": lambda: L.pea_affinity_bwd(ctypes.byref(desc), P(E), None, P(G), P(one), P(dE), None, st),
           "inf": lambda: L.pea_affinity_infer(ctypes.byref(desc), P(E), None, P(affs), st)}
    vox = B * Z * Y * X
    out = []
    for kn, fn in fns.items():
        for _ in range(2): assert fn() == 0
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): assert fn() == 0
        b.record(); b.synchronize()
        us = a.elapsed_time(b) / 5 * 1e3
        ab = {"fwd": 64 + 12 * K, "bwd": 128 + 8 * K, "inf": 64 + 4 * K}[kn]   # 3D has no mask
        out.append("%s %9.1f us (%4.0f GB/s alg)" % (kn, us, ab * vox / us / 1e3))
    print("%-44s %s" % (name, "  ".join(out)), flush=True)
    del E, T, Wt, affs, G, dE
    torch.cuda.empty_cache()
