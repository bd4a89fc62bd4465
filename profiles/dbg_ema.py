import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
g = dict(np.load("tests/golden/g2d_ema_detach.npz"))
dev = torch.device("cuda:0")
cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
offs = g["offsets"].tolist()
def run():
    loss, affs = pkg.ema_embedding_loss(cu(g["e"]), cu(g["ema"]), cu(g["target"]), cu(g["weight"]), cu(g["mask"]), pkg.WeightedMSE(), offs, affs0_weight=float(g["affs0_weight"]))
    return loss.item(), affs.cpu().numpy()
l1, a1 = run()
os.environ["PEA_FWD_V"] = "0"
l0, a0 = run()
d = np.abs(a1 - a0)
print("loss", l1, l0, float(g["loss"]), "max diff", d.max(), "ref diff old", np.abs(a0 - g["affs"]).max())
idx = np.argwhere(d > 1e-4)
print(len(idx), "bad; first", idx[:12].tolist())
import collections
print("by channel", collections.Counter(idx[:,1].tolist()))
print("rows", sorted(set(idx[:,2].tolist()))[:40]); print("cols", sorted(set(idx[:,3].tolist()))[:60])
for (b,c,y,x) in idx[:6].tolist() + idx[60:64].tolist() + idx[-4:].tolist():
    v = a1[b,c,y,x]
    w = np.argwhere(np.abs(a0 - v) < 1e-6)
    print((b,c,y,x), "got", v, "want", a0[b,c,y,x], "matches elsewhere:", w[:4].tolist())
