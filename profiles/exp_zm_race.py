#!/usr/bin/env python3
"""bit-reproducibility screen of the march kernels on the sub-volume: forward outputs (affs, g, 1/norm, loss) and the gradient over
repeated launches, under the switches named in SW (comma-separated NAME=VALUE|NAME=VALUE sets)"""
import ctypes, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
aff = importlib.import_module(ge.PKG_NAME + ".utils.affinity_ours")
dev = torch.device("cuda:0")
op, L = pkg.affinity_op, pkg._lib.lib()
P = lambda x: None if x is None else ctypes.c_void_p(x.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
offs = aff.axis_offsets_3d(aff.NORM5_SHIFTS)
if os.environ.get("STENCIL") == "n26":
    offs = [[dz, dy, dx] for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1) if (dz, dy, dx) != (0, 0, 0)]
Z, Y, X = (int(v) for v in os.environ.get("DIMS", "24,1024,1024").split(","))
B, K = 1, len(offs)
g = torch.Generator(device=dev); g.manual_seed(1)
E = torch.randn(B, 16, Z, Y, X, device=dev, generator=g)
T = (torch.rand(B, K, Z, Y, X, device=dev, generator=g) < 0.7).float()
Wt = torch.rand(B, K, Z, Y, X, device=dev, generator=g) + 0.5
spec = op.AffinitySpec(3, offs, None, pkg._lib.BORDER_CROP_ZERO, pkg._lib.NORM_CROPPED)
desc = op.make_desc(spec, E)
wsb = L.pea_workspace_bytes(ctypes.byref(desc)); work = torch.empty(max(wsb, 4) // 4, device=dev)
assert L.pea_workspace_init(ctypes.c_void_p(work.data_ptr()), wsb, None) == 0
one = torch.ones((), device=dev)


def fwd():
    affs = torch.empty(B, K, Z, Y, X, device=dev); G = torch.empty_like(affs); lossv = torch.empty(1 + K, device=dev)
    INV = torch.empty(B, 1, Z, Y, X, device=dev)
    assert L.pea_affinity_fwd_ex(ctypes.byref(desc), P(E), None, P(T), P(Wt), None, P(affs), P(G), P(INV), P(lossv), P(work), wsb, st) == 0
    return affs, G, INV, lossv


def bwd(affs, G, INV):
    dE = torch.empty_like(E)
    assert L.pea_affinity_bwd_ex2(ctypes.byref(desc), P(E), None, P(G), P(INV), P(affs), P(one), P(dE), None, st) == 0
    return dE


for sw in os.environ.get("SW", "").split("|"):
    sets = [kv.split("=") for kv in sw.split(",") if kv]
    for k, v in sets: pkg._lib.set_switch(k, v)
    ref = fwd()
    bad_f = [0, 0, 0, 0]
    for _ in range(int(os.environ.get("N", "6"))):
        cur = fwd()
        for i in range(4):
            bad_f[i] += int((ref[i] != cur[i]).sum().item())
    dref = bwd(*ref[:3])
    bad_b, where = 0, None
    for _ in range(int(os.environ.get("N", "6"))):
        d = bwd(*ref[:3])
        ne = (dref != d)
        n = int(ne.sum().item())
        if n and where is None:
            idx = ne.nonzero()[:8].tolist()
            where = idx
        bad_b += n
    print("%-40s fwd mismatches affs/g/inv/loss %s   bwd mismatches %d %s" % (sw or "(default)", bad_f, bad_b, where or ""), flush=True)
    for k, v in sets: pkg._lib.set_switch(k, None)
