#!/usr/bin/env python3
"""the 3D EMA cross loss (ema_embedding_loss_norm5, scripts_ac3ac4/main.py:224) and the whole 3D loss section on the sub-volume:
what the reference's 3D training loop calls per step, beside the self loss bench.py --config c4 times"""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
dev = torch.device("cuda:0")
Z, Y, X = (int(v) for v in os.environ.get("DIMS", "24,1024,1024").split(","))
g = torch.Generator(device=dev); g.manual_seed(1)
E = torch.randn(1, 16, Z, Y, X, device=dev, generator=g).requires_grad_(True)
EMA = (E.detach() + 0.3 * torch.randn(1, 16, Z, Y, X, device=dev, generator=g))
T = (torch.rand(1, 12, Z, Y, X, device=dev, generator=g) < 0.7).float()
W = torch.rand(1, 12, Z, Y, X, device=dev, generator=g) + 0.5
crit = pkg.WeightedMSE()


def timed(fn, n=4):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


def self_step():
    E.grad = None
    loss, _ = pkg.embedding_loss_norm5(E, T, W, crit); loss.backward()


def ema_step():
    E.grad = None
    loss, _ = pkg.ema_embedding_loss_norm5(E, EMA, T, W, crit); loss.backward()


print("self norm5 fwd+bwd  %8.2f ms" % timed(self_step), flush=True)
print("ema  norm5 fwd+bwd  %8.2f ms" % timed(ema_step), flush=True)
