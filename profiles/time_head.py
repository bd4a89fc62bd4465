#!/usr/bin/env python3
"""The embedding head (1x1 convolution C -> D) at the CVPPP bench shape: pea_head_fwd / pea_head_bwd against torch's own
GPU convolution (MIOpen / rocBLAS) forward and backward, HIP events.  Algorithmic bytes: fwd 4(C+D), bwd 4(2C+D) per px."""
import ctypes, os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
L = pkg._lib.lib()
dev = torch.device("cuda:0")
P = lambda x: ctypes.c_void_p(x.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def timed(fn, n=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / n * 1e3


for (B, C, D, sp) in ((8, 32, 16, (544, 544)), (8, 64, 16, (272, 272)), (8, 32, 32, (544, 544)), (2, 28, 16, (18, 160, 160))):
    g = torch.Generator(device=dev); g.manual_seed(1)
    x = torch.randn((B, C) + sp, device=dev, generator=g)
    w = torch.randn((D, C) + (1,) * len(sp), device=dev, generator=g) * 0.2
    b = torch.randn(D, device=dev, generator=g)
    de = torch.randn((B, D) + sp, device=dev, generator=g)
    S = x[0, 0].numel()
    e = torch.empty((B, D) + sp, device=dev); dx = torch.empty_like(x); dW = torch.empty(D, C, device=dev); db = torch.empty(D, device=dev)
    wsb = L.pea_head_workspace_bytes(C, D); work = torch.empty(wsb // 4, device=dev)
    w2 = w.reshape(D, C).contiguous()
    t_f = timed(lambda: L.pea_head_fwd(B, C, D, S, P(x), P(w2), P(b), P(e), st))
    t_b = timed(lambda: L.pea_head_bwd(B, C, D, S, P(x), P(w2), P(de), P(dx), P(dW), P(db), P(work), wsb, st))
    t_w = timed(lambda: L.pea_head_bwd(B, C, D, S, P(x), P(w2), P(de), None, P(dW), P(db), P(work), wsb, st))
    conv = F.conv3d if len(sp) == 3 else F.conv2d
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    t_tf = timed(lambda: conv(xr, wr, br))
    out = conv(xr, wr, br)
    t_tb = timed(lambda: torch.autograd.grad(out, (xr, wr, br), de, retain_graph=True))
    px = B * S
    print("B=%d C=%d D=%d %-14s  pea fwd %7.1f us (%4.0f GB/s)  bwd %7.1f us (%4.0f GB/s; dW+db alone %6.1f us)   torch fwd %7.1f us  bwd %7.1f us"
          % (B, C, D, "x".join(map(str, sp)), t_f, 4 * (C + D) * px / t_f / 1e3, t_b, 4 * (2 * C + D) * px / t_b / 1e3, t_w, t_tf, t_tb), flush=True)
