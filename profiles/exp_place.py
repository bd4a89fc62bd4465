#!/usr/bin/env python3
"""Is the D = 16 cross backward's slow state (104-114 us instead of 94-101, per process) a matter of WHERE its tensors sit in memory?
One process, the headline shape: the tensors are freed and allocated afresh (with dummy allocations of random sizes in between so
that the allocator hands out other pages), the backward timed on each placement."""
import ctypes, importlib, os, random, sys
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
spec = op.AffinitySpec(2, offsets, None, pkg._lib.BORDER_CIRCULAR, pkg._lib.NORM_BX)
random.seed(int(os.environ.get("SEED", "1")))
keep = []
for trial in range(int(os.environ.get("TRIALS", "12"))):
    g0 = torch.Generator(device=dev); g0.manual_seed(1)
    E = torch.randn(B, D, H, W, device=dev, generator=g0)
    T = (torch.rand(B, K, H, W, device=dev, generator=g0) < 0.7).float()
    Wt = torch.rand(B, K, H, W, device=dev, generator=g0) + 0.5
    M = torch.ones(B, K, H, W, device=dev, dtype=torch.uint8)
    desc = op.make_desc(spec, E)
    affs = torch.empty(B, K, H, W, device=dev); lossv = torch.empty(1 + K, device=dev); INV = torch.empty(B, 1, H, W, device=dev)
    G = torch.empty(B, K, H, W, device=dev); dE = torch.empty_like(E); one = torch.ones((), device=dev)
    wsb = L.pea_workspace_bytes(ctypes.byref(desc)); work = torch.empty(max(wsb, 4) // 4, device=dev)
    assert L.pea_workspace_init(ctypes.c_void_p(work.data_ptr()), wsb, None) == 0
    fw = lambda: L.pea_affinity_fwd_ex(ctypes.byref(desc), P(E), None, P(T), P(Wt), P(M), P(affs), P(G), P(INV), P(lossv), P(work), wsb, st)
    bw = lambda: L.pea_affinity_bwd_ex2(ctypes.byref(desc), P(E), None, P(G), P(INV), P(affs), P(one), P(dE), None, st)
    assert fw() == 0
    res = []
    for fn in (fw, bw):
        for _ in range(20 if trial else 300): assert fn() == 0   # (the first trial also warms the clocks up)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(40): assert fn() == 0
        b.record(); b.synchronize()
        res.append(a.elapsed_time(b) / 40 * 1e3)
    # the step as bench.py runs it: forward and backward alternating
    for _ in range(10): fw(); bw()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(40): fw(); bw()
    b.record(); b.synchronize()
    print("trial %2d  e %x g %x de %x   fwd %.1f us  bwd %.1f us  alternating step %.1f us" % (trial, E.data_ptr(), G.data_ptr(), dE.data_ptr(), res[0], res[1],
          a.elapsed_time(b) / 40 * 1e3), flush=True)
    del E, T, Wt, M, affs, lossv, INV, G, dE, work
    keep.append(torch.empty(random.randrange(1, 64) << 20, dtype=torch.uint8, device=dev))  # shifts what the next round gets
    if len(keep) > 6: keep.pop(0)
    torch.cuda.empty_cache()
