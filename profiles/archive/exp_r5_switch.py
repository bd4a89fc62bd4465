#!/usr/bin/env python3
"""A/B of PEA_* switches inside ONE process (the step's time differs by process): bench.py's step
(embedding_loss forward + pea.backward, B x 16 x 544^2, K=10) in batches of 100 steps, the variants alternating round by round.
    python profiles/exp_r5_switch.py "PEA_BWD_REV=1" "PEA_BWD_W3=1" "PEA_BWD_REV=1,PEA_BWD_W3=1"      (the empty set is always variant 0)
Prints wall ms/step per batch and the in-step HIP-event durations of the forward and of the backward (C ABI)."""
import os, sys, time, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
import importlib
pkg = ge.load_package()
synth = importlib.import_module(ge.PKG_NAME + ".utils.synth")
dev = torch.device("cuda:0")
B = int(os.environ.get("EXP_B", "8"))
D, H, W = 16, 544, 544
offsets = pkg.multi_offset([1, 3, 5, 9, 27], 4)
e, t, w, m = synth.synth_inputs_2d(B, D, H, W, offsets, seed=555)
E = torch.from_numpy(e).to(dev).requires_grad_(True)
T, Wt, M = torch.from_numpy(t).to(dev), torch.from_numpy(w).to(dev), torch.from_numpy(m).to(dev)
crit = pkg.WeightedMSE()
variants = [dict()] + [dict(kv.split("=") for kv in a.split(",") if kv) for a in sys.argv[1:]]
names = set(k for v in variants for k in v)


def select(v):
    for k in names:
        if k in v:
            os.environ[k] = v[k]
        else:
            os.environ.pop(k, None)
    pkg._lib.reload_env()


def step():
    E.grad = None
    loss, affs, _ = pkg.embedding_loss(E, T, Wt, M, crit, offsets)
    pkg.backward(loss)
    return loss


def in_step(iters=50):
    op, L = pkg.affinity_op, pkg._lib.lib()
    K = len(offsets)
    spec = op.AffinitySpec(2, offsets, None, pkg._lib.BORDER_CIRCULAR, pkg._lib.NORM_BX)
    Ed = E.detach()
    desc = op.make_desc(spec, Ed)
    affs, lossv, G = torch.empty(B, K, H, W, device=dev), torch.empty(1 + K, device=dev), torch.empty(B, K, H, W, device=dev)
    wsb = L.pea_workspace_bytes(ctypes.byref(desc))
    work = torch.empty(max(wsb, 4) // 4, device=dev)
    assert L.pea_workspace_init(ctypes.c_void_p(work.data_ptr()), wsb, None) == 0
    dE, one, INV = torch.empty_like(Ed), torch.ones((), device=dev), torch.empty(B, H, W, device=dev)
    P = lambda x: ctypes.c_void_p(x.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    fwd = lambda: L.pea_affinity_fwd_ex(ctypes.byref(desc), P(Ed), None, P(T), P(Wt), P(M), P(affs), P(G), P(INV), P(lossv), P(work), wsb, st)
    bwd = lambda: L.pea_affinity_bwd_ex2(ctypes.byref(desc), P(Ed), None, P(G), P(INV), P(affs), P(one), P(dE), None, st)
    s = torch.cuda.current_stream()
    for _ in range(10):
        fwd(); bwd()
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(iters)]
    for a, b, c in ev:
        a.record(s); fwd(); b.record(s); bwd(); c.record(s)
    ev[-1][2].synchronize()
    return (sum(a.elapsed_time(b) for a, b, _ in ev) / iters * 1e3, sum(b.elapsed_time(c) for _, b, c in ev) / iters * 1e3, dE.clone())


for _ in range(300):
    step()
torch.cuda.synchronize()
ref = None
for rnd in range(int(os.environ.get("EXP_ROUNDS", "4"))):
    for vi, v in enumerate(variants):
        select(v)
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 10
        f_us, b_us, dE = in_step()
        if ref is None:
            ref = dE
        same = bool(torch.equal(ref, dE))
        print("round %d  %-40s  %.4f ms/step   fwd %.1f us  bwd %.1f us   grad bit-equal to variant 0: %s"
              % (rnd, ",".join("%s=%s" % kv for kv in v.items()) or "(default)", ms, f_us, b_us, same), flush=True)
