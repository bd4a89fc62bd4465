#!/usr/bin/env python3
"""time fwd / bwd / inf at a D = 32 shape (BBBC-like: B=8 x 32 x 544 x 544, shifts 1,3,5,9,11 x neighbor 4); D=64 with the env
variable D; honours PEA_* env"""
import ctypes, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
synth = importlib.import_module(ge.PKG_NAME + ".utils.synth")
B, D, H, W = 8, int(os.environ.get("D", 32)), 544, 544
dev = torch.device("cuda:0")
op, L = pkg.affinity_op, pkg._lib.lib()
P = lambda x: ctypes.c_void_p(x.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
offsets = pkg.multi_offset([1, 3, 5, 9, 11], 4)
K = len(offsets)
e, t, w, m = synth.synth_inputs_2d(B, D, H, W, offsets, 555)
E, T, Wt, M = (torch.from_numpy(x).to(dev) for x in (e, t, w, m))
if os.environ.get("F16", "0") == "1":  # f16 storage of the embedding / its gradient, f32 accumulation
    E = E.half()
desc = op.make_desc(op.AffinitySpec(2, offsets, None, pkg._lib.BORDER_CIRCULAR, pkg._lib.NORM_BX), E)
affs = torch.empty(B, K, H, W, device=dev); G = torch.empty(B, K, H, W, device=dev); lossv = torch.empty(1 + K, device=dev)
wsb = L.pea_workspace_bytes(ctypes.byref(desc)); work = torch.empty(max(wsb, 4) // 4, device=dev)
assert L.pea_workspace_init(ctypes.c_void_p(work.data_ptr()), wsb, None) == 0  # the loss-state block: prepared once
dE = torch.empty_like(E); one = torch.ones((), device=dev)
fns = {"fwd": lambda: L.pea_affinity_fwd(ctypes.byref(desc), P(E), None, P(T), P(Wt), P(M), P(affs), P(G), P(lossv), P(work), wsb, st),
       "bwd": lambda: L.pea_affinity_bwd(ctypes.byref(desc), P(E), None, P(G), P(one), P(dE), None, st),
       "inf": lambda: L.pea_affinity_infer(ctypes.byref(desc), P(E), None, P(affs), st)}
LAB = torch.from_numpy(synth.synth_labels(B, (1, H, W), 555)[:, 0].copy()).to(dev)
WTAB = torch.empty(B * K * 2, device=dev)
CNTB = L.pea_targets_workspace_bytes(ctypes.byref(desc)); CNT = torch.empty(CNTB // 4, dtype=torch.int32, device=dev)
assert L.pea_label_weights(ctypes.byref(desc), P(LAB), 5, P(WTAB), P(CNT), CNTB, st) == 0
if D in (16, 32):
    fns["labels_step"] = lambda: L.pea_affinity_fwd_bwd_labels(ctypes.byref(desc), P(E), None, P(LAB), P(WTAB), 5, P(affs), P(lossv), None, P(dE), P(work), wsb, st)
fns["fwd"]()
for name, fn in fns.items():
    for _ in range(5): fn()
    ts = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): assert fn() == 0
        b.record(); b.synchronize()
        ts.append(a.elapsed_time(b) / 20 * 1e3)
    print("D=%d%s %s min %.1f us" % (D, " f16" if E.dtype == torch.float16 else "", name, min(ts)), flush=True)
