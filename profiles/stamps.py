#!/usr/bin/env python3
"""Diagnostic: per-section s_memtime stamps of the phased backward (wave 0, second tile of each workgroup)."""
import os, sys, ctypes
sys.argv = [sys.argv[0], "bwd", "0"]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "one_kernel.py")).read().split("if which ==")[0])
import numpy as np
buf = torch.zeros(256 * 64, dtype=torch.int64, device=dev)
fns["fwd"]()
for _ in range(3): fns["bwd"]()
torch.cuda.synchronize()
L.pea_debug_stamps.argtypes = [ctypes.c_void_p]
L.pea_debug_stamps(ctypes.c_void_p(buf.data_ptr()))
fns["bwd"]()
torch.cuda.synchronize()
L.pea_debug_stamps(None)
a = buf.cpu().numpy().reshape(256, 64)
n = int((a[0] != 0).sum())
d = np.diff(a[:, :n], axis=1).astype(np.float64)
names = ["write", "bar1", "issue", "compute", "epi", "bar2", "gcopy", "loop"]
print("stamps per wg:", n)
med = np.median(d, axis=0)
for i, v in enumerate(med):
    print("%2d %-8s %8.0f ticks (p10 %6.0f p90 %6.0f)" % (i, names[i % 8], v, np.percentile(d[:, i], 10), np.percentile(d[:, i], 90)))
print("total", med.sum(), "ticks for one tile; 100 MHz ticks => %.2f us" % (med.sum() / 100.0))
