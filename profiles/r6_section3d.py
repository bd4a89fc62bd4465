#!/usr/bin/env python3
"""The 3D training loop's loss section (scripts_ac3ac4/main.py:219-237: embedding_loss_norm5 + ema_embedding_loss_norm5 + four
embedding_loss_norm1 on the deep-supervision heads + backward + border fill + relu) at the shape the reference trains on
(ac3ac4.yaml:52 batch 2; data_provider_labeled_deep.py:53 crops of 18 x 160 x 160), a few calls of each form, for a rocprofv3
--kernel-trace run (profiles/r6_section3d_timeline.sh).   python profiles/r6_section3d.py [one_node|finished|composed] [calls]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
which = sys.argv[1] if len(sys.argv) > 1 else "one_node"
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda:0")
B, Z, Y, X = 2, 18, 160, 160
g = torch.Generator(device=dev).manual_seed(900)
crit = pkg.WeightedMSE()
emb = torch.randn(B, 16, Z, Y, X, generator=g, device=dev)
ema = torch.randn(B, 16, Z, Y, X, generator=g, device=dev)
emds = [torch.randn(B, 16, Z, Y >> j, X >> j, generator=g, device=dev) for j in (4, 3, 2, 1)]
target = (torch.rand(B, 12, Z, Y, X, generator=g, device=dev) < 0.6).float()
weight = torch.rand(B, 12, Z, Y, X, generator=g, device=dev) + 0.5
downs = [torch.cat([(torch.rand(B, 3, Z, Y >> j, X >> j, generator=g, device=dev) < 0.6).float(),
                    torch.rand(B, 3, Z, Y >> j, X >> j, generator=g, device=dev) + 0.5], dim=1) for j in (1, 2, 3, 4)]


def run():
    xs = [emb.detach().requires_grad_(True)] + [e.detach().requires_grad_(True) for e in emds]
    if which == "composed":
        loss, pred = pkg.ac3ac4_loss_section_composed(xs[0], xs[1:], ema, target, weight, downs, crit, embedding_mode=5)
    else:
        loss, pred = pkg.ac3ac4_loss_section(xs[0], xs[1:], ema, target, weight, downs, crit, embedding_mode=5, finish_pred=which == "finished")
    loss.backward()
    if which != "finished":
        pkg.finish_pred_3d_(pred)


for _ in range(calls):
    run()
    torch.cuda.synchronize()   # one call per burst: the trace shows a call's kernels without the next call's behind them
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10):
    run()
b.record(); b.synchronize()
print("%s: %.1f us per call (10 calls back to back)" % (which, a.elapsed_time(b) * 100), flush=True)
