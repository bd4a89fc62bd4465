#!/usr/bin/env python3
"""What does a plain streaming kernel reach at the forward's read / write mix?  torch elementwise kernels over 47 M-float tensors:
c = a + b (2 reads : 1 write, 564 MB), copy (1 : 1), sum-like read-only (a.sum()), fill (write only).  The forward moves
365 MB in + 199 MB out (1.83 : 1)."""
import torch
dev = torch.device("cuda:0")
n = 47 * 1024 * 1024
a, b, c = (torch.randn(n, device=dev) for _ in range(3))
d = torch.randn(n, device=dev)
def timed(fn, it=50):
    for _ in range(10): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / it * 1e3
nb = n * 4 / 1e6
for name, fn, mb in (("add  c=a+b   (2R:1W)", lambda: torch.add(a, b, out=c), 3 * nb),
                     ("copy c=a     (1R:1W)", lambda: c.copy_(a), 2 * nb),
                     ("addcmul d=a+b*c (3R:1W)", lambda: torch.addcmul(a, b, c, out=d), 4 * nb),
                     ("fill          (0R:1W)", lambda: c.fill_(1.0), nb),
                     ("sum           (1R:0W)", lambda: a.sum(), nb)):
    t = timed(fn)
    print("%-26s %7.1f us  %6.0f MB  %5.2f TB/s" % (name, t, mb, mb / t), flush=True)
