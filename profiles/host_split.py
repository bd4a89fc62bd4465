#!/usr/bin/env python3
"""host time of the two halves of a step on a tiny problem (host-bound): embedding_loss(...) and loss.backward()"""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
synth = importlib.import_module(ge.PKG_NAME + ".utils.synth")
dev = torch.device("cuda:0")
offsets = pkg.multi_offset([1, 3, 5, 9, 27], 4)
e, t, w, m = synth.synth_inputs_2d(1, 16, 64, 96, offsets, 555)
E = torch.from_numpy(e).to(dev).requires_grad_(True)
T, Wt, M = (torch.from_numpy(x).to(dev) for x in (t, w, m))
crit = pkg.WeightedMSE()
N = 3000
for rep in range(2):
    tf = tb = 0.0
    for _ in range(N):
        E.grad = None
        t0 = time.perf_counter()
        loss, affs, _ = pkg.embedding_loss(E, T, Wt, M, crit, offsets)
        t1 = time.perf_counter()
        loss.backward()
        t2 = time.perf_counter()
        tf += t1 - t0; tb += t2 - t1
    torch.cuda.synchronize()
print("forward call %.1f us   backward call %.1f us" % (tf / N * 1e6, tb / N * 1e6))
# the same through the C ABI alone (no autograd, no allocations)
import ctypes
op, L = pkg.affinity_op, pkg._lib.lib()
Ed = E.detach()
desc = op.make_desc(op.AffinitySpec(2, offsets, None, pkg._lib.BORDER_CIRCULAR, pkg._lib.NORM_BX), Ed)
K = len(offsets)
affs = torch.empty(1, K, 64, 96, device=dev); G = torch.empty_like(affs); lossv = torch.empty(1 + K, device=dev)
wsb = L.pea_workspace_bytes(ctypes.byref(desc)); work = torch.empty(max(wsb, 4) // 4, device=dev)
assert L.pea_workspace_init(ctypes.c_void_p(work.data_ptr()), wsb, None) == 0  # the loss-state block: prepared once
INV = torch.empty(1, 64, 96, device=dev); dE = torch.empty_like(Ed); one = torch.ones((), device=dev)
P = lambda x: ctypes.c_void_p(x.data_ptr())
st = op._stream()
t0 = time.perf_counter()
for _ in range(N):
    L.pea_affinity_fwd_ex(ctypes.byref(desc), P(Ed), None, P(T), P(Wt), P(M), P(affs), P(G), P(INV), P(lossv), P(work), wsb, st)
    L.pea_affinity_bwd_ex(ctypes.byref(desc), P(Ed), None, P(G), P(INV), P(one), P(dE), None, st)
torch.cuda.synchronize()
print("C ABI pair through ctypes: %.1f us per step" % ((time.perf_counter() - t0) / N * 1e6))
