#!/usr/bin/env python3
"""Round 6: the march kernels' super-block walk (PEA_ZM_SUP, csrc/pea_xdma.h march_tile) on BASELINE configs[3] (16 x 24 x 1024^2, norm5),
same process, same buffers: forward and backward INSIDE the alternating step (fwd, bwd, fwd, bwd ...) per (block, super-block) shape.
usage: python profiles/r6_sup.py [STENCIL=n26 for the box march backward]"""
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
SUPVARS = ["PEA_ZM_SUP"]
if os.environ.get("STENCIL") == "n26":  # the 26-neighbourhood: k_fwd_box (tile per plane) and k_bwd_boxm (march), their walks by PEA_BOX_SUP / PEA_BOXM_SUP
    offs = [[dz, dy, dx] for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1) if (dz, dy, dx) != (0, 0, 0)]
    SUPVARS = ["PEA_BOX_SUP", "PEA_BOXM_SUP"]
Z, Y, X = (int(v) for v in os.environ.get("DIMS", "24,1024,1024").split(","))
B, K = int(os.environ.get("B", "1")), len(offs)
iters = int(os.environ.get("ITERS", "8"))
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
fwd = lambda: L.pea_affinity_fwd_ex(ctypes.byref(desc), P(E), None, P(T), P(Wt), None, P(affs), P(G), P(INV), P(lossv), P(work), wsb, st)
bwd = lambda: L.pea_affinity_bwd_ex2(ctypes.byref(desc), P(E), None, P(G), P(INV), P(affs), P(one), P(dE), None, st)


def in_step(n):
    s = torch.cuda.current_stream()
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(n)]
    for a, b, c in ev:
        a.record(s); assert fwd() == 0; b.record(s); assert bwd() == 0; c.record(s)
    ev[-1][2].synchronize()
    return sum(a.elapsed_time(b) for a, b, _ in ev) / n, sum(b.elapsed_time(c) for _, b, c in ev) / n


ref = None
in_step(3)
for rnd in range(int(os.environ.get("ROUNDS", "2"))):
    for blk in os.environ.get("BLOCKS", "16x2,8x4,4x8").split(","):
        gy, gx = (int(v) for v in blk.split("x"))
        for sup in os.environ.get("SUPS", "0,8,4,2,1").split(","):
            pkg._lib.set_switch("PEA_ZBLK_Y", gy); pkg._lib.set_switch("PEA_ZBLK_X", gx); [pkg._lib.set_switch(v, sup) for v in SUPVARS]
            in_step(2)
            f, b = in_step(iters)
            chk = (float(lossv[0]), float(dE.double().abs().sum()))
            ref = ref or chk
            print("round %d  block %2d x %d  sup_x %s  fwd %7.1f us  bwd %7.1f us  step %7.1f us  %s"
                  % (rnd, gy, gx, sup, f * 1e3, b * 1e3, (f + b) * 1e3, "same bits" if chk == ref else "DIFFERENT RESULT %r vs %r" % (chk, ref)), flush=True)
