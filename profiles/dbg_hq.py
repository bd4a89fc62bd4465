import ctypes, sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
import __graft_entry__ as ge
pkg = ge.load_package(); orc = ge.load_oracle()
import importlib
synth = importlib.import_module(ge.PKG_NAME + ".utils.synth")
dev = torch.device("cuda:0")
op = pkg.affinity_op
offsets = pkg.multi_offset([1, 3, 5, 9, 27], 4)[:8]
B, D, H, W = 2, 64, 72, 104
e, t, w, m = synth.synth_inputs_2d(B, D, H, W, offsets, 90 + D)
e = e.astype(np.float16).astype(np.float32)
spec = op.AffinitySpec(2, offsets, None, pkg._lib.BORDER_CIRCULAR, pkg._lib.NORM_BX)
def run():
    et = torch.from_numpy(e).to(dev).half().requires_grad_(True)
    loss, affs, _ = op.FusedAffinityMSE.apply(et, None, torch.from_numpy(t).to(dev), torch.from_numpy(w).to(dev), torch.from_numpy(m).to(dev), spec)
    (loss * 0.75).backward()
    return et.grad.float().cpu().numpy()
res = {}
for hw in ("2", "1"):
    pkg._lib.set_switch("PEA_H16_HW", hw); res[hw] = run()
a, b = res["2"], res["1"]
d = a != b
print("differ", d.sum(), "of", d.size, "max abs", np.abs(a - b).max(), "max rel", (np.abs(a - b) / (np.abs(b) + 1e-30))[d].max() if d.any() else 0)
idx = np.argwhere(d)[:10]
for i in idx: print(tuple(i), a[tuple(i)], b[tuple(i)])
print("by x:", np.bincount(np.argwhere(d)[:, 3], minlength=W)[:], sep="\n")
print("by y:", np.bincount(np.argwhere(d)[:, 2], minlength=H)[:], sep="\n")
