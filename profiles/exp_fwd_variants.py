#!/usr/bin/env python3
"""Round 3: the forward's two new switches, A/B in one process at the bench shape (B=8 x 16 x 544^2, K=10):
PEA_FWD_WG3 (three workgroups per CU, 7.5 KB planes, 80 VGPRs) x PEA_LOSS_TICKET (loss finished inside the forward kernel vs a
second tiny launch).  Times are HIP-event durations inside the alternating fwd / bwd step, as bench.py's kernel_ms."""
import ctypes, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
import bench
pkg = ge.load_package()
synth = importlib.import_module(ge.PKG_NAME + ".utils.synth")
dev = torch.device("cuda:0")
op, L = pkg.affinity_op, pkg._lib.lib()
B, D, H, W = int(os.environ.get("B", 8)), int(os.environ.get("D", 16)), int(os.environ.get("HW", 544)), int(os.environ.get("HW", 544))
shifts = [int(v) for v in os.environ.get("SHIFTS", "1,3,5,9,27").split(",")]
offsets = pkg.multi_offset(shifts, 4)
K = len(offsets)
g = torch.Generator(device=dev).manual_seed(555)
E = torch.randn(B, D, H, W, generator=g, device=dev)
T = (torch.rand(B, K, H, W, generator=g, device=dev) < 0.6).float()
Wt = torch.rand(B, K, H, W, generator=g, device=dev) + 0.5
M = (torch.rand(B, K, H, W, generator=g, device=dev) < 0.9).to(torch.uint8)
spec = op.AffinitySpec(2, offsets, None, pkg._lib.BORDER_CIRCULAR, pkg._lib.NORM_BX)
desc = op.make_desc(spec, E)
affs, G = torch.empty(B, K, H, W, device=dev), torch.empty(B, K, H, W, device=dev)
lossv, dE, INV = torch.empty(1 + K, device=dev), torch.empty_like(E), torch.empty(B, H, W, device=dev)
work, wsb = op.workspace(dev, desc)
P = lambda x: ctypes.c_void_p(x.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
fwd = lambda: L.pea_affinity_fwd_ex(ctypes.byref(desc), P(E), None, P(T), P(Wt), P(M), P(affs), P(G), P(INV), P(lossv), P(work), wsb, st)
bwd = lambda: L.pea_affinity_bwd_ex(ctypes.byref(desc), P(E), None, P(G), P(INV), None, P(dE), None, st)
for _ in range(200):
    fwd(); bwd()
torch.cuda.synchronize()
ref = None
for rnd in range(2):
    for wg3 in ("1", "0"):
        for ticket in ("1", "0"):
            pkg._lib.set_switch("PEA_FWD_WG3", wg3)
            pkg._lib.set_switch("PEA_LOSS_TICKET", ticket)
            assert fwd() == 0
            bench.in_step_times_ms(fwd, bwd, 20)
            kf, kb = bench.in_step_times_ms(fwd, bwd, 100)
            lv = lossv.cpu().numpy().copy()
            if ref is None:
                ref = lv
            same = bool((lv == ref).all())
            print("round %d  WG3=%s TICKET=%s  fwd %.1f us  bwd %.1f us  sum %.1f us   loss %.6f bit-identical %s" % (rnd, wg3, ticket, kf * 1e3, kb * 1e3, (kf + kb) * 1e3, lv[0], same), flush=True)
