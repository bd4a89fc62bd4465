#!/usr/bin/env python3
"""The headline step's two states (0.200 / 0.215 ms per step; the backward 94-98 / 104-110 us): do they change inside a process?
bench.py's step (embedding_loss forward + pea.backward, B=8 x 16 x 544^2, K=10) in batches of 100, wall time per batch, 3000 steps;
PRE=n: n back-to-back steps without any host synchronisation first."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
import importlib
pkg = ge.load_package()
synth = importlib.import_module(ge.PKG_NAME + ".utils.synth")
dev = torch.device("cuda:0")
B, D, H, W = 8, 16, 544, 544
offsets = pkg.multi_offset([1, 3, 5, 9, 27], 4)
e, t, w, m = synth.synth_inputs_2d(B, D, H, W, offsets, seed=555)
E = torch.from_numpy(e).to(dev).requires_grad_(True)
T, Wt, M = torch.from_numpy(t).to(dev), torch.from_numpy(w).to(dev), torch.from_numpy(m).to(dev)
crit = pkg.WeightedMSE()


def step():
    E.grad = None
    loss, affs, _ = pkg.embedding_loss(E, T, Wt, M, crit, offsets)
    pkg.backward(loss)


for _ in range(int(os.environ.get("PRE", "0"))):
    step()
torch.cuda.synchronize()
out = []
for b in range(30):
    t0 = time.perf_counter()
    for _ in range(100):
        step()
    torch.cuda.synchronize()
    out.append((time.perf_counter() - t0) * 10)  # ms per step
print("PRE=%s  ms/step per 100-step batch: %s" % (os.environ.get("PRE", "0"), " ".join("%.4f" % v for v in out)), flush=True)
