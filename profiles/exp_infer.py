#!/usr/bin/env python3
"""A/B: inference (pea_affinity_infer, affs only) on k_fwd_tiled (default) vs the LDS-DMA forward at three workgroups per CU
(PEA_INFER_XDMA=1), B=8 x 16 x 544^2, K=10; max |difference| of the two maps."""
import ctypes, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
synth = importlib.import_module(ge.PKG_NAME + ".utils.synth")
dev = torch.device("cuda:0")
op, L = pkg.affinity_op, pkg._lib.lib()
B, D, H, W = 8, int(os.environ.get("D", 16)), int(os.environ.get("HW", 544)), int(os.environ.get("HW", 544))
offsets = pkg.multi_offset([int(v) for v in os.environ.get("SHIFTS", "1,3,5,9,27").split(",")], 4)[:int(os.environ.get("K", 10))]
K = len(offsets)
E = torch.from_numpy(synth.synth_embedding((B, D, H, W), 555)).to(dev)
if os.environ.get("F16"):   # f16 storage (BASELINE configs[4])
    E = E.half()
desc = op.make_desc(op.AffinitySpec(2, offsets, None, pkg._lib.BORDER_CIRCULAR, pkg._lib.NORM_BX), E)
P = lambda x: ctypes.c_void_p(x.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
outs = {}
for sw in ("0", "1", "0", "1"):
    pkg._lib.set_switch("PEA_INFER_XDMA", sw)
    affs = torch.empty(B, K, H, W, device=dev)
    fn = lambda: L.pea_affinity_infer(ctypes.byref(desc), P(E), None, P(affs), st)
    assert fn() == 0
    for _ in range(30): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(100): fn()
    b.record(); b.synchronize()
    outs[sw] = affs
    print("D=%d %dx%d K=%d  PEA_INFER_XDMA=%s  %.1f us" % (D, H, W, K, sw, a.elapsed_time(b) * 10), flush=True)
print("max |tiled - cross| = %.2e" % float((outs["0"] - outs["1"]).abs().max()))
