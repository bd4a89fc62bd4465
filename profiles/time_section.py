#!/usr/bin/env python3
"""The loss section of the CVPPP training loop (scripts_cvppp/main.py:284-311: five self losses over the scales
544..34 + the EMA cross loss, then loss.backward()) at B=8, D=16: tensor path (targets / weights / masks resident),
tensor path with the targets generated on the GPU each step, and the labels-in path.  HIP events, whole section."""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
synth = importlib.import_module(ge.PKG_NAME + ".utils.synth")
dev = torch.device("cuda:0")
B, D, H, W, nb_half = 8, int(os.environ.get("D", 16)), 544, 544, 2
f16 = os.environ.get("F16", "0") == "1"
offsets = pkg.multi_offset([1, 3, 5, 9, 27] if D == 16 else [1, 3, 5, 9, 11], 4)  # cvppp.yaml / bbbc039v1.yaml shifts
crit = pkg.WeightedMSE()
lab = synth.synth_labels(B, (1, H, W), 555)[:, 0]
labs = [torch.from_numpy(np.ascontiguousarray(lab[:, ::2 ** j, ::2 ** j])).to(dev) for j in range(5)]
dt = torch.float16 if f16 else torch.float32
emb = [torch.from_numpy(synth.synth_embedding((B, D, H >> j, W >> j), 600 + j)).to(dev).to(dt) for j in range(5)]
ema = torch.from_numpy(synth.synth_embedding((B, D, H, W), 700)).to(dev).to(dt)


def targets():
    tt, mm, ww = pkg.gen_targets(labs[0], offsets, padding=True)
    downs = []
    for j in range(1, 5):
        k = nb_half * (5 - j)
        tj, mj, wj = pkg.gen_targets(labs[j], offsets[:k], padding=True)
        downs.append(torch.cat([tj, wj, mj.float()], dim=1))
    return tt, mm, ww, downs


TT, MM, WW, DOWNS = targets()


def leaves():
    return [e.detach().requires_grad_(True) for e in emb]


def tensor_section(gen):
    tt, mm, ww, downs = targets() if gen else (TT, MM, WW, DOWNS)
    x = leaves()
    loss, pred, _ = pkg.cvppp_loss_section(x[0], x[1:], ema, tt, ww, mm, downs, crit, offsets, nb_half, relu_pred=True)
    loss.backward()  # (pred is already relu'd: the reference's next statement rides on the kernel's store)


def labels_section():
    x = leaves()
    loss, pred, _ = pkg.cvppp_loss_section_from_labels(x[0], x[1:], ema, labs[0], labs[1:], crit, offsets, nb_half, relu_pred=True)
    loss.backward()


def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        b.record(); b.synchronize()
        ts.append(a.elapsed_time(b) / n * 1e3)
    return min(ts)


def graphed(fn):
    """capture fn (forward + backward + epilogue) in a HIP graph; returns the replay callable"""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return g.replay


TABS = pkg.cvppp_label_weight_tables(labs[0], labs[1:], offsets, nb_half)


def labels_section_tables():
    x = leaves()
    loss, pred, _ = pkg.cvppp_loss_section_from_labels(x[0], x[1:], ema, labs[0], labs[1:], crit, offsets, nb_half, relu_pred=True,
                                                       weight_tables=TABS)
    loss.backward()


px = B * H * W
if os.environ.get("ONLY"):   # one case only, for a kernel trace (profiles/prof_script.sh): ONLY=tensor | labels | tables
    fn = {"tensor": lambda: tensor_section(False), "labels": labels_section, "tables": labels_section_tables}[os.environ["ONLY"]]
    for _ in range(40): fn()
    torch.cuda.synchronize()
    print("%s: %.1f us per section" % (os.environ["ONLY"], timed(fn)))
    sys.exit(0)
for _ in range(30):  # clocks and allocator pools settle before the first timed case
    tensor_section(False)
torch.cuda.synchronize()
for name, fn in (("tensor path, targets resident", lambda: tensor_section(False)),
                 ("tensor path + gen_targets each step", lambda: tensor_section(True)),
                 ("labels-in path", labels_section),
                 ("labels-in path, weight tables computed ahead", labels_section_tables)):
    us = timed(fn)
    ug = timed(graphed(fn))
    print("%s D=%d loss section (%s): eager %8.1f us, HIP-graph replay %8.1f us = %6.0f Mpx/s of full-resolution pixels"
          % ("f16" if f16 else "f32", D, name, us, ug, px / ug), flush=True)
