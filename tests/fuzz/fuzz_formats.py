#!/usr/bin/env python3
"""Randomised sweep over the data-format paths either side of the loss:
  (a) pea_gen_targets against the numpy restatement of gen_affs_ours / weight_binary_ratio (bit-exact: integer / byte work),
  (b) the 3D labels-in losses (norm1 / norm5, self and EMA) against gen_targets + the tensor functions,
  (c) the replicate-border variant (norm6) against the C oracle,
  (d) the device stitcher against the reference's numpy statements (bit-exact).
usage: fuzz_formats.py [cases] [seed]; exits non-zero on a disagreement.  The oracle is the checker (test infrastructure)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge
pkg = ge.load_package()
orc = ge.load_oracle(); orc.build()
dev = torch.device("cuda:0")
ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
crit = pkg.WeightedMSE()
bad = 0


def fail(tag, ctx, msg):
    global bad
    bad += 1
    print("MISMATCH %s %s: %s" % (tag, ctx, msg), flush=True)


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


for it in range(ncase):
    g = torch.Generator(device=dev); g.manual_seed(900 + it)
    # ---- (a) target generation, 2D and 3D, both flag sets
    three_d = rng.random() < 0.4
    B = int(rng.integers(1, 3))
    dims = [int(rng.integers(2, 6)), int(rng.integers(5, 40)), int(rng.integers(5, 60))] if three_d else [int(rng.integers(4, 70)), int(rng.integers(4, 90))]
    K = int(rng.integers(1, 9))
    offs = [[int(rng.integers(-min(d - 1, 9), min(d - 1, 9) + 1)) for d in dims] for _ in range(K)]
    offs = [o if any(o) else o[:-1] + [1] for o in offs]
    lab = torch.randint(0, 4, [B] + dims, device=dev, generator=g).int()
    padding, both = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    t, m, w = pkg.gen_targets(lab, offs, padding=padding, both_foreground=both)
    lab4 = lab.cpu().numpy().reshape([B] + ([1] if not three_d else []) + dims)
    o3 = [([0] * (3 - len(o))) + o for o in offs]
    t_np, m_np = orc.np_gen_targets(lab4, o3, padding=padding, both_foreground=both)
    w_np = orc.np_weight_binary_ratio(t_np)
    ctx = "case %d dims=%s offs=%s pad=%d fg=%d" % (it, dims, offs, padding, both)
    if not np.array_equal(t.cpu().numpy().reshape(t_np.shape), t_np) or not np.array_equal(m.cpu().numpy().reshape(m_np.shape), m_np):
        fail("gen_targets", ctx, "target / mask differ")
    elif w_np is not None and not np.array_equal(w.cpu().numpy().reshape(w_np.shape), w_np):
        fail("gen_targets", ctx, "weights differ (max %.3e)" % np.abs(w.cpu().numpy().reshape(w_np.shape) - w_np).max())

    # ---- (b) 3D labels-in losses against targets + tensor functions
    which = int(rng.integers(0, 3))
    Z, Y, X = int(rng.integers(2 if which == 0 else 5, 8)), int(rng.integers(30, 70)), int(rng.integers(30, 80))  # norm5 reaches 4 planes / 27 px
    lab3 = torch.randint(0, 4, (1, Z, Y // 6 + 1, X // 6 + 1), device=dev, generator=g).repeat_interleave(6, 2).repeat_interleave(6, 3)[:, :, :Y, :X].contiguous().int()
    e3 = torch.randn(1, 16, Z, Y, X, device=dev, generator=g)
    ema3 = torch.randn(1, 16, Z, Y, X, device=dev, generator=g)
    aw = float(rng.choice([1.0, 2.0]))
    sh = orc.norm_offsets([1, 1, 1] if which == 0 else [1, 1, 1, 2, 3, 3, 3, 9, 9, 4, 27, 27])
    t3, _, w3 = pkg.gen_targets(lab3, sh, padding=False, both_foreground=True, want_mask=False)
    outs = []
    for path in ("tensor", "labels"):
        et = e3.clone().requires_grad_(True)
        if which == 0:
            r = pkg.embedding_loss_norm1(et, t3, w3, crit, affs0_weight=aw) if path == "tensor" else pkg.embedding_loss_norm1_from_labels(et, lab3, crit, affs0_weight=aw)
        elif which == 1:
            r = pkg.embedding_loss_norm5(et, t3, w3, crit, affs0_weight=aw) if path == "tensor" else pkg.embedding_loss_norm5_from_labels(et, lab3, crit, affs0_weight=aw)
        else:
            r = pkg.ema_embedding_loss_norm5(et, ema3, t3, w3, crit, affs0_weight=aw) if path == "tensor" else \
                pkg.ema_embedding_loss_norm5_from_labels(et, ema3, lab3, crit, affs0_weight=aw)
        (r[0] * 0.5).backward()
        outs.append((r[0].item(), r[1].cpu().numpy(), et.grad.cpu().numpy()))
    ctx = "case %d which=%d Z=%d %dx%d aw=%g" % (it, which, Z, Y, X, aw)
    if abs(outs[0][0] - outs[1][0]) > 1e-5 * abs(outs[0][0]) or np.abs(outs[0][1] - outs[1][1]).max() > 1e-5 or rel(outs[1][2], outs[0][2]) > 1e-4:
        fail("labels3d", ctx, "loss %.6f / %.6f  affs %.2e  grad %.2e" % (outs[0][0], outs[1][0], np.abs(outs[0][1] - outs[1][1]).max(), rel(outs[1][2], outs[0][2])))

    # ---- (c) replicate border (norm6) against the C oracle
    if it % 2 == 0:
        Zr, Yr, Xr = int(rng.integers(2, 6)), int(rng.integers(6, 24)), int(rng.integers(6, 30))
        Kr = int(rng.integers(1, 7))
        offr = [[int(rng.integers(-2, 3)), int(rng.integers(-6, 7)), int(rng.integers(-6, 7))] for _ in range(Kr)]
        offr = [[max(-(d - 1), min(d - 1, v)) for v, d in zip(o, (Zr, Yr, Xr))] for o in offr]
        offr = [o if any(o) else [0, 0, 1] for o in offr]
        er = np.random.default_rng(1000 + it).standard_normal((1, 16, Zr, Yr, Xr)).astype(np.float32)
        tr = (np.random.default_rng(2000 + it).random((1, Kr, Zr, Yr, Xr)) < 0.6).astype(np.float32)
        wr = (np.random.default_rng(3000 + it).random((1, Kr, Zr, Yr, Xr)) + 0.5).astype(np.float32)
        et = torch.from_numpy(er).to(dev).requires_grad_(True)
        loss, affs = pkg.embedding_loss_norm6(et, torch.from_numpy(tr).to(dev), torch.from_numpy(wr).to(dev), crit, shift=offr)
        loss.backward()
        dd = orc.desc_3d_replicate(er, offr)
        o_affs, o_loss = orc.c_fwd(dd, er, None, tr, wr, None)
        o_grad, _ = orc.c_bwd(dd, er, None, tr, wr, None)
        ctx = "case %d dims=%s offs=%s" % (it, (Zr, Yr, Xr), offr)
        if np.abs(affs.cpu().numpy() - o_affs).max() > 1e-5 or abs(loss.item() - o_loss[0]) > 1e-5 * abs(o_loss[0]) or rel(et.grad.cpu().numpy(), o_grad) > 1e-4:
            fail("replicate", ctx, "affs %.2e loss %.6f / %.6f grad %.2e" % (np.abs(affs.cpu().numpy() - o_affs).max(), loss.item(), o_loss[0], rel(et.grad.cpu().numpy(), o_grad)))

    # ---- (d) stitcher against the numpy statements of provider_valid.py:320-349
    if it % 4 == 0:
        C = 3
        vol_shape = (int(rng.integers(6, 12)), int(rng.integers(20, 40)), int(rng.integers(20, 40)))
        out_size = (int(rng.integers(2, 5)), int(rng.integers(8, 16)), int(rng.integers(8, 16)))
        st = pkg.VolumeStitcher(C, vol_shape, out_size, dev)
        ref_out = np.zeros((C,) + vol_shape, np.float32); ref_w = np.zeros(vol_shape, np.float32)
        wv = st.weight_vol.cpu().numpy().reshape(out_size)
        for _ in range(5):
            pos = [int(rng.integers(0, vol_shape[a] - out_size[a] + 1)) for a in range(3)]
            v = torch.rand((C,) + out_size, device=dev, generator=g)
            st.add_vol(v, pos)
            sl = tuple(slice(pos[a], pos[a] + out_size[a]) for a in range(3))
            ref_out[(slice(None),) + sl] += v.cpu().numpy() * wv
            ref_w[sl] += wv
        got = st.out_affs.cpu().numpy()
        if not np.array_equal(got, ref_out) or not np.array_equal(st.weight_map.cpu().numpy().reshape(ref_w.shape), ref_w):
            fail("stitch", "case %d vol=%s out=%s" % (it, vol_shape, out_size), "accumulators differ (max %.3e)" % np.abs(got - ref_out).max())

print("fuzz_formats: %d cases, %d mismatches" % (ncase, bad))
sys.exit(1 if bad else 0)
