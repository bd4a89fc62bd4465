#!/usr/bin/env python3
"""Does the D = 16 cross backward's two-state duration (94-98 or 104-110 us, per process) follow the RELATIVE placement of its
tensors?  One process, the headline shape; e fixed, g / de carved out of a pool at a sweep of byte offsets."""
import ctypes, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
dev = torch.device("cuda:0")
op, L = pkg.affinity_op, pkg._lib.lib()
P = lambda x: None if x is None else ctypes.c_void_p(x.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
B, D, H, W = 8, 16, 544, 544
offsets = pkg.multi_offset([1, 3, 5, 9, 27], 4)
K = len(offsets)
g0 = torch.Generator(device=dev); g0.manual_seed(1)
E = torch.randn(B, D, H, W, device=dev, generator=g0)
T = (torch.rand(B, K, H, W, device=dev, generator=g0) < 0.7).float()
Wt = torch.rand(B, K, H, W, device=dev, generator=g0) + 0.5
M = torch.ones(B, K, H, W, device=dev, dtype=torch.uint8)
spec = op.AffinitySpec(2, offsets, None, pkg._lib.BORDER_CIRCULAR, pkg._lib.NORM_BX)
desc = op.make_desc(spec, E)
affs = torch.empty(B, K, H, W, device=dev); lossv = torch.empty(1 + K, device=dev)
INV = torch.empty(B, 1, H, W, device=dev)
wsb = L.pea_workspace_bytes(ctypes.byref(desc)); work = torch.empty(max(wsb, 4) // 4, device=dev)
assert L.pea_workspace_init(ctypes.c_void_p(work.data_ptr()), wsb, None) == 0
one = torch.ones((), device=dev)
nG, nE = B * K * H * W, B * D * H * W
pool = torch.empty(nG + nE + (64 << 20) // 4, device=dev)  # g, de and 64 MB of slack
print("addresses: e %x  pool %x  affs %x" % (E.data_ptr(), pool.data_ptr(), affs.data_ptr()))


def run(off_g, off_d, iters=30):
    G = pool[off_g // 4: off_g // 4 + nG].view(B, K, H, W)
    dE = pool[(32 << 20) // 4 + nG + off_d // 4: (32 << 20) // 4 + nG + off_d // 4 + nE].view(B, D, H, W)
    assert L.pea_affinity_fwd_ex(ctypes.byref(desc), P(E), None, P(T), P(Wt), P(M), P(affs), P(G), P(INV), P(lossv), P(work), wsb, st) == 0
    bw = lambda: L.pea_affinity_bwd_ex2(ctypes.byref(desc), P(E), None, P(G), P(INV), P(affs), P(one), P(dE), None, st)
    for _ in range(5): assert bw() == 0
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): assert bw() == 0
    b.record(); b.synchronize()
    return a.elapsed_time(b) / iters * 1e3


for rep in range(2):
    for off in (0, 256, 1024, 4096, 16384, 65536, 262144, 1 << 20, (1 << 20) + 4096, 3 << 20, 5 << 20 | 8192):
        print("rep %d  de offset %8d: %6.1f us   g offset %8d: %6.1f us" % (rep, off, run(0, off), off, run(off, 0)), flush=True)
