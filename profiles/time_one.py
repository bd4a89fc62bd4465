#!/usr/bin/env python3
"""time one entry point (fwd|bwd|inf) at the bench shape with HIP events; honours PEA_* env"""
import os, sys, subprocess
sys.argv = [sys.argv[0], sys.argv[1], "0"]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "one_kernel.py")).read().split("if which ==")[0])
fn = fns[which]
for _ in range(10): fn()
torch.cuda.synchronize()
ts = []
for rnd in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(30): fn()
    b.record(); b.synchronize()
    ts.append(a.elapsed_time(b) / 30 * 1e3)
print("%s min %.1f us med %.1f us" % (which, min(ts), sorted(ts)[2]))
