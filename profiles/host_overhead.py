#!/usr/bin/env python3
"""How much of a bench step is host time?  Times the Python loop (launches only) and the drained loop."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
synth = importlib.import_module(ge.PKG_NAME + ".utils.synth")
dev = torch.device("cuda:0")
offsets = pkg.multi_offset([1, 3, 5, 9, 27], 4)
B, D, H, W = 8, 16, 544, 544
e, t, w, m = synth.synth_inputs_2d(B, D, H, W, offsets, 555)
E = torch.from_numpy(e).to(dev).requires_grad_(True)
T, Wt, M = (torch.from_numpy(x).to(dev) for x in (t, w, m))
crit = pkg.WeightedMSE()
def step():
    E.grad = None
    loss, affs, _ = pkg.embedding_loss(E, T, Wt, M, crit, offsets)
    loss.backward()
for _ in range(3 if os.environ.get("PEA_STEPS") else 20): step()
torch.cuda.synchronize()
for n in ((int(os.environ["PEA_STEPS"]),) if os.environ.get("PEA_STEPS") else (50, 200)):
    t0 = time.perf_counter()
    for _ in range(n): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("n=%d host loop %.1f us/step, drained %.1f us/step" % (n, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
if os.environ.get("PEA_NO_TINY"):
    sys.exit(0)
# host cost alone: same loop on a tiny problem (kernels ~ microseconds)
e2, t2_, w2, m2 = synth.synth_inputs_2d(1, D, 64, 64, offsets[:8], 555)
E2 = torch.from_numpy(e2).to(dev).requires_grad_(True)
T2, W2, M2 = (torch.from_numpy(x).to(dev) for x in (t2_, w2, m2))
def step2():
    E2.grad = None
    loss, affs, _ = pkg.embedding_loss(E2, T2, W2, M2, crit, offsets[:8])
    loss.backward()
for _ in range(20): step2()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): step2()
torch.cuda.synchronize()
print("tiny problem: %.1f us/step (host-bound floor of the Python path)" % ((time.perf_counter() - t0) / 200 * 1e6))
