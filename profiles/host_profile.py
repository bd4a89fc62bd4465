#!/usr/bin/env python3
"""cProfile of the Python path of one bench step (embedding_loss + backward) on a tiny problem: where the host's ~156 us go"""
import cProfile, importlib, os, pstats, sys
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
def step():
    E.grad = None
    loss, affs, _ = pkg.embedding_loss(E, T, Wt, M, crit, offsets)
    loss.backward()
for _ in range(200): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(2000): step()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)
