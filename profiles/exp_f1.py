import ctypes, os, sys, torch
sys.path.insert(0, '/root/repo')
import __graft_entry__ as ge
pkg = ge.load_package()
dev = torch.device("cuda:0"); op, L = pkg.affinity_op, pkg._lib.lib()
P = lambda x: ctypes.c_void_p(x.data_ptr()) if x is not None else None
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
B, D, H, W, HC = 8, 16, 544, 544, 32
offsets = pkg.multi_offset([1, 3, 5, 9, 27], neighbor=4); K = len(offsets)
E = torch.randn(B, D, H, W, device=dev); T = (torch.rand(B, K, H, W, device=dev) < 0.6).float(); Wt = torch.rand(B, K, H, W, device=dev) + 0.5
M = (torch.rand(B, K, H, W, device=dev) < 0.9).to(torch.uint8)
desc = op.make_desc(op.AffinitySpec(2, offsets, None, pkg._lib.BORDER_CIRCULAR, pkg._lib.NORM_BX), E)
affs = torch.empty(B, K, H, W, device=dev); G = torch.empty_like(affs); lossv = torch.empty(1 + K, device=dev)
wsb = L.pea_workspace_bytes(ctypes.byref(desc)); work = torch.empty(max(wsb, 4) // 4, device=dev)
INV = torch.empty(B, H, W, device=dev); dE = torch.empty_like(E); one = torch.ones((), device=dev)
hx = torch.randn(B, HC, H, W, device=dev); hw = torch.randn(D, HC, device=dev) * 0.2
hdx, hdw, hdb = torch.empty_like(hx), torch.empty(D, HC, device=dev), torch.empty(D, device=dev)
fb = L.pea_bwd_head_workspace_bytes(ctypes.byref(desc), HC); fwork = torch.empty(max(fb, 4) // 4, device=dev)
hws = L.pea_head_workspace_bytes(HC, D); hwork = torch.empty(hws // 4, device=dev)
assert L.pea_affinity_fwd_ex(ctypes.byref(desc), P(E), None, P(T), P(Wt), P(M), P(affs), P(G), P(INV), P(lossv), P(work), wsb, st) == 0
def t(fn, n=30):
    for _ in range(5): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); b.synchronize(); return a.elapsed_time(b) / n * 1e3
bwd = lambda: L.pea_affinity_bwd_ex(ctypes.byref(desc), P(E), None, P(G), P(INV), P(one), P(dE), None, st)
hb = lambda: L.pea_head_bwd(B, HC, D, H * W, P(hx), P(hw), P(dE), P(hdx), P(hdw), P(hdb), P(hwork), hws, st)
fused = lambda de=None: L.pea_affinity_bwd_head(ctypes.byref(desc), P(E), P(G), P(INV), P(one), None, P(hx), P(hw), HC, P(hdx), P(hdw), P(hdb), P(de), P(fwork), fb, st)
print("bwd %.1f  head_bwd %.1f" % (t(bwd), t(hb)))
# (profiles/r2c_f1_fused_backward.txt also has the run with the epilogue's dx / dW halves switched off by a debug build)
print("fused %.1f   fused+de %.1f" % (t(fused), t(lambda: fused(dE))))
