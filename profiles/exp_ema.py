#!/usr/bin/env python3
"""ema_embedding_loss (detached second operand) at B=8 x 16 x 544^2, K=10: forward / role-A backward on the cross kernels against
the tiled kernels (PEA_FWD_XDMA=0 / PEA_BWD_XDMA=0), through the C ABI, HIP events"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
dev = torch.device("cuda:0"); op, L = pkg.affinity_op, pkg._lib.lib()
P = lambda x: ctypes.c_void_p(x.data_ptr()) if x is not None else None
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
B, D, H, W = 8, 16, 544, 544
offsets = pkg.multi_offset([1, 3, 5, 9, 27], neighbor=4); K = len(offsets)
E = torch.randn(B, D, H, W, device=dev); EO = torch.randn(B, D, H, W, device=dev)
T = (torch.rand(B, K, H, W, device=dev) < 0.6).float(); Wt = torch.rand(B, K, H, W, device=dev) + 0.5
M = (torch.rand(B, K, H, W, device=dev) < 0.9).to(torch.uint8)
desc = op.make_desc(op.AffinitySpec(2, offsets, None, pkg._lib.BORDER_CIRCULAR, pkg._lib.NORM_BX), E)
import copy
dacc = copy.copy(desc); dacc.flags |= pkg._lib.FLAG_ACCUMULATE_DE
affs = torch.empty(B, K, H, W, device=dev); G = torch.empty_like(affs); lossv = torch.empty(1 + K, device=dev)
wsb = L.pea_workspace_bytes(ctypes.byref(desc)); work = torch.empty(max(wsb, 4) // 4, device=dev)
assert L.pea_workspace_init(ctypes.c_void_p(work.data_ptr()), wsb, None) == 0  # the loss-state block: prepared once
INV2 = torch.empty(2, B, H, W, device=dev); dE = torch.zeros_like(E); one = torch.ones((), device=dev)
def t(fn, n=30):
    for _ in range(5):
        rc = fn()
        assert rc == 0, rc
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); b.synchronize(); return a.elapsed_time(b) / n * 1e3
fwd = lambda inv: L.pea_affinity_fwd_ex(ctypes.byref(desc), P(E), P(EO), P(T), P(Wt), P(M), P(affs), P(G), P(inv), P(lossv), P(work), wsb, st)
bwd = lambda d, inv: L.pea_affinity_bwd_ex(ctypes.byref(d), P(E), P(EO), P(G), P(inv), P(one), P(dE), None, st)
for _ in range(300): fwd(INV2)   # clocks settle
torch.cuda.synchronize()
print("cross : fwd %.1f  bwd %.1f  bwd accumulate %.1f" % (t(lambda: fwd(INV2)), t(lambda: bwd(desc, INV2)), t(lambda: bwd(dacc, INV2))))
print("tiled : fwd %.1f  bwd %.1f" % (t(lambda: fwd(None)), t(lambda: bwd(desc, None))))
# the pair's backward in one launch: cross kernel with the second phase against the tiled two-phase kernel
G0 = torch.empty_like(G); INV0 = torch.empty(B, H, W, device=dev)
assert L.pea_affinity_fwd_ex(ctypes.byref(desc), P(E), None, P(T), P(Wt), P(M), P(affs), P(G0), P(INV0), P(lossv), P(work), wsb, st) == 0
assert fwd(INV2) == 0
dual = lambda a, b: L.pea_affinity_bwd_dual_ex(ctypes.byref(desc), P(E), P(EO), P(G0), P(G), P(a), P(b), P(one), P(one), P(dE), st)
selfb = lambda: L.pea_affinity_bwd_ex(ctypes.byref(desc), P(E), None, P(G0), P(INV0), P(one), P(dE), None, st)
print("pair  : cross dual %.1f  tiled dual %.1f  (self backward alone %.1f)" % (t(lambda: dual(INV0, INV2[1])), t(lambda: dual(None, None)), t(selfb)))
