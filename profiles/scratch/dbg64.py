import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge
pkg = ge.load_package(); op, L = pkg.affinity_op, pkg._lib.lib()
dev = torch.device("cuda:0")
B, D, dims = 1, 64, [1, 44, 199]
offs = [[0, 6, -5], [0, -26, 25], [0, 0, -12], [0, -27, -23], [0, 3, 16]]
K = len(offs)
for f16 in (False, True):
  for ema in (False, True):
    for use_m in (False, True):
        g = torch.Generator(device=dev); g.manual_seed(3)
        dt = torch.float16 if f16 else torch.float32
        e = torch.randn([B, D] + dims, device=dev, generator=g).to(dt)
        o = torch.randn([B, D] + dims, device=dev, generator=g).to(dt) if ema else None
        t = (torch.rand([B, K] + dims, device=dev, generator=g) < 0.6).float()
        w = torch.rand([B, K] + dims, device=dev, generator=g) + 0.5
        m = (torch.rand([B, K] + dims, device=dev, generator=g) < 0.9).to(torch.uint8) if use_m else None
        spec = op.AffinitySpec(3, offs, [1.0] * K, 0, 1)
        d = op.make_desc(spec, e)
        p = lambda x: ctypes.c_void_p(x.data_ptr()) if x is not None else None
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        wsb = L.pea_workspace_bytes(ctypes.byref(d)); work = torch.empty(wsb // 4 + 1, device=dev)
        out = []
        for direct in ("1", "0"):
            os.environ["PEA_FORCE_DIRECT"] = direct
            affs = torch.zeros([B, K] + dims, device=dev); gg = torch.zeros_like(affs); rows = torch.zeros(1 + K, device=dev)
            rc = L.pea_affinity_fwd(ctypes.byref(d), p(e), p(o), p(t), p(w), p(m), p(affs), p(gg), p(rows), p(work), wsb, st)
            torch.cuda.synchronize()
            out.append((rc, affs, gg, rows))
        da = (out[0][1] - out[1][1]).abs().amax(dim=(0, 2, 3, 4)).cpu().numpy()
        dg = (out[0][2] - out[1][2]).abs().amax(dim=(0, 2, 3, 4)).cpu().numpy()
        print("f16=%d ema=%d mask=%d rc=%s  per-channel |d affs| %s  |d g| %s  rows direct %s tiled %s" % (
            f16, ema, use_m, (out[0][0], out[1][0]), np.array2string(da, precision=1), np.array2string(dg, precision=2),
            np.array2string(out[0][3].cpu().numpy(), precision=3), np.array2string(out[1][3].cpu().numpy(), precision=3)))
