#!/usr/bin/env python3
"""the z-march kernels on the AC3/AC4 sub-volume (16 x 24 x 1024^2, norm5): forward as a whole and with outputs removed
(no g; no g and no affs; inference), backward.  CASES=fwd,bwd,... selects; for rocprofv3 --pmc passes use ITERS=3."""
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
if os.environ.get("STENCIL") == "norm1":
    offs = [[-1, 0, 0], [0, -1, 0], [0, 0, -1]]
if os.environ.get("STENCIL") == "n26":
    offs = [[dz, dy, dx] for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1) if (dz, dy, dx) != (0, 0, 0)]
Z, Y, X = (int(v) for v in os.environ.get("DIMS", "24,1024,1024").split(","))
B, K = int(os.environ.get("B", "1")), len(offs)
iters = int(os.environ.get("ITERS", "6"))
g = torch.Generator(device=dev); g.manual_seed(1)
E = torch.randn(B, 16, Z, Y, X, device=dev, generator=g)
T = (torch.rand(B, K, Z, Y, X, device=dev, generator=g) < 0.7).float()
Wt = torch.rand(B, K, Z, Y, X, device=dev, generator=g) + 0.5
spec = op.AffinitySpec(3, offs, None, pkg._lib.BORDER_CROP_ZERO, pkg._lib.NORM_CROPPED)
desc = op.make_desc(spec, E)
affs = torch.empty(B, K, Z, Y, X, device=dev); G = torch.empty_like(affs); lossv = torch.empty(1 + K, device=dev)
INV = torch.empty(B, 1, Z, Y, X, device=dev)
wsb = L.pea_workspace_bytes(ctypes.byref(desc)); work = torch.empty(max(wsb, 4) // 4, device=dev)
assert L.pea_workspace_init(ctypes.c_void_p(work.data_ptr()), wsb, None) == 0
dE = torch.empty_like(E); one = torch.ones((), device=dev)
fwd = lambda a, gg: L.pea_affinity_fwd_ex(ctypes.byref(desc), P(E), None, P(T), P(Wt), None, P(a), P(gg), P(INV), P(lossv), P(work), wsb, st)
fns = {"fwd": lambda: fwd(affs, G), "fwd_nog": lambda: fwd(affs, None), "fwd_noout": lambda: fwd(None, None),
       "infer": lambda: L.pea_affinity_infer(ctypes.byref(desc), P(E), None, P(affs), st),
       "bwd": lambda: L.pea_affinity_bwd_ex2(ctypes.byref(desc), P(E), None, P(G), P(INV), P(affs), P(one), P(dE), None, st)}
assert fns["fwd"]() == 0


def run(kn, tag=""):
    fn = fns[kn]
    for _ in range(2): assert fn() == 0
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): assert fn() == 0
    b.record(); b.synchronize()
    print("%-10s %8.1f us %s" % (kn, a.elapsed_time(b) / iters * 1e3, tag), flush=True)
    if kn in ("fwd_nog", "fwd_noout"): assert fns["fwd"]() == 0  # the backward reads g and affs


for kn in os.environ.get("CASES", "fwd,fwd_nog,fwd_noout,infer,bwd").split(","):
    run(kn)
for tag in [t for t in os.environ.get("VARIANTS", "").split(",") if t]:  # diagnostic builds (profiles/build_variant.sh)
    lm = pkg._lib
    keep = (lm.SO_PATH, lm._lib)
    lm.SO_PATH, lm._lib = os.path.join(os.path.dirname(keep[0]), "libpea_hip_%s.so" % tag), None
    L = lm.lib()
    assert L.pea_workspace_init(ctypes.c_void_p(work.data_ptr()), wsb, None) == 0
    for kn in os.environ.get("VCASES", "fwd,infer,bwd").split(","):
        run(kn, "(variant %s)" % tag)
    lm.SO_PATH, lm._lib = keep
    L = keep[1]
if os.environ.get("ILV"):  # (needs the PEA_XCD_INTERLEAVE switch of the experiment's build: not in the product)
    for gy, gx in [tuple(int(v) for v in b.split('x')) for b in os.environ.get('BLOCKS', '16x2,8x4,4x8,4x4').split(',')]:
        for ilv in ("0", "1"):
            pkg._lib.set_switch("PEA_ZBLK_Y", gy); pkg._lib.set_switch("PEA_ZBLK_X", gx); pkg._lib.set_switch("PEA_XCD_INTERLEAVE", ilv)
            for kn in os.environ.get("CASES", "fwd,bwd").split(","):
                run(kn, "(block %d x %d, interleave %s)" % (gy, gx, ilv))
    sys.exit(0)
if os.environ.get("STENCIL") == "n26":
    pkg._lib.set_switch("PEA_BOXM", "0"); run("bwd", "(PEA_BOXM=0: the per-(z, tile) box backward)")
    pkg._lib.set_switch("PEA_BOXM", None); run("bwd", "(marching again)")
    for gy, gx in [(8, 4), (4, 8), (16, 4), (32, 1), (4, 4)]:
        pkg._lib.set_switch("PEA_ZBLK_Y", gy); pkg._lib.set_switch("PEA_ZBLK_X", gx)
        run("bwd", "(block %d x %d)" % (gy, gx))
    pkg._lib.set_switch("PEA_ZBLK_Y", None); pkg._lib.set_switch("PEA_ZBLK_X", None)
    sys.exit(0)
if os.environ.get("AB", "1") == "1":  # the same box, the same buffers: the alternatives
    for gy, gx in [tuple(int(v) for v in b.split('x')) for b in os.environ.get('BLOCKS', '4x8,16x2,2x16,32x1,1x32,4x4,8x8,16x4').split(',')]:  # the block of tile columns an XCD marches
        pkg._lib.set_switch("PEA_ZBLK_Y", gy); pkg._lib.set_switch("PEA_ZBLK_X", gx)
        run("fwd", "(block %d x %d tile columns)" % (gy, gx)); run("bwd", "(block %d x %d)" % (gy, gx))
    pkg._lib.set_switch("PEA_ZBLK_Y", None); pkg._lib.set_switch("PEA_ZBLK_X", None)
    pkg._lib.set_switch("PEA_ZMARCH", "0"); run("fwd", "(PEA_ZMARCH=0: tile-per-plane cross kernels)"); run("bwd", "(PEA_ZMARCH=0)")
    pkg._lib.set_switch("PEA_ZMARCH", None)
