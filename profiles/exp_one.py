#!/usr/bin/env python3
"""Run one backward variant a few times (for rocprofv3 --pmc passes): python profiles/exp_one.py <old|x0|x1|...> [iters]"""
import ctypes, os, sys
sys.argv = [sys.argv[0]] + sys.argv[1:]
which = sys.argv[1]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "exp_cross.py")).read().split("def timeit")[0]
sys.argv = [sys.argv[0], "8"]
exec(src)
for _ in range(iters):
    if which == "old":
        assert L.pea_affinity_bwd(ctypes.byref(desc), P(E), None, P(G), P(one), P(dE), None, st) == 0
    else:
        assert X.pea_x_bwd(ctypes.byref(desc), P(E), P(INV), P(G), P(one), P(dE2), int(which[1:]), st) == 0
torch.cuda.synchronize()
