#!/usr/bin/env python3
"""The training loop's loss section (scripts_cvppp/main.py:284-312: five self losses over the scales + the EMA cross loss + backward
+ relu) at B = 8 x 544^2, a few calls of each form, for a rocprofv3 --kernel-trace run (profiles/r5_section_timeline.sh): which
kernel runs when, on which stream.   python profiles/r5_section.py [one_node|labels|composed] [calls]"""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
synth = importlib.import_module(ge.PKG_NAME + ".utils.synth")
which = sys.argv[1] if len(sys.argv) > 1 else "one_node"
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda:0")
B, D, H, W = 8, 16, 544, 544
offsets = pkg.multi_offset([1, 3, 5, 9, 27], 4)
nb_half = 2
crit = pkg.WeightedMSE()
labs = [torch.from_numpy(np.ascontiguousarray(synth.synth_labels(B, (1, H, W), 555)[:, 0][:, ::2 ** j, ::2 ** j])).to(dev) for j in range(5)]
embs = [torch.from_numpy(synth.synth_embedding((B, D, H >> j, W >> j), 600 + j)).to(dev) for j in range(5)]
ema = torch.from_numpy(synth.synth_embedding((B, D, H, W), 700)).to(dev)
tt, mm, ww = pkg.gen_targets(labs[0], offsets, padding=True)
downs = []
for j in range(1, 5):
    k = nb_half * (5 - j)
    tj, mj, wj = pkg.gen_targets(labs[j], offsets[:k], padding=True)
    downs.append(torch.cat([tj, wj, mj.float()], dim=1))


def run():
    x = [e.detach().requires_grad_(True) for e in embs]
    if which == "labels":
        loss, pred, _ = pkg.cvppp_loss_section_from_labels(x[0], x[1:], ema, labs[0], labs[1:], crit, offsets, nb_half, relu_pred=True)
    elif which == "one_node":
        loss, pred, _ = pkg.cvppp_loss_section(x[0], x[1:], ema, tt, ww, mm, downs, crit, offsets, nb_half, relu_pred=True)
    else:
        loss, pred, _ = pkg.cvppp_loss_section_composed(x[0], x[1:], ema, tt, ww, mm, downs, crit, offsets, nb_half)
    loss.backward()
    if which == "composed":
        pkg.finish_pred_2d_(pred)


for _ in range(calls):
    run()
    torch.cuda.synchronize()   # one call per burst: the trace shows a call's kernels without the next call's behind them
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10):
    run()
b.record(); b.synchronize()
print("%s: %.1f us per call (10 calls back to back)" % (which, a.elapsed_time(b) * 100), flush=True)
