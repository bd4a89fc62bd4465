#!/usr/bin/env python3
"""Randomised agreement sweep over the alternative paths to the same numbers:
  (a) the labels-in step (embedding_loss_from_labels / ema_...) against gen_targets + the tensor path,
  (b) the LDS-DMA cross kernels against the tiled kernels (PEA_FWD_XDMA / PEA_BWD_XDMA = 0) on the tensor path,
  (c) the embedding head against torch's GPU convolution (every supported channel pair, ragged pixel counts),
  (d) the six-loss section as one autograd node and from labels against its call-by-call composition.
usage: fuzz_paths.py [cases] [seed]; exits non-zero on a disagreement."""
import os, sys
import numpy as np, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge
pkg = ge.load_package()
dev = torch.device("cuda:0")
ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
crit = pkg.WeightedMSE()
bad = 0
worst = {}


def rel(a, b):
    return float((a.float() - b.float()).abs().max() / b.float().abs().max().clamp_min(1e-30))


def note(tag, ok, ctx, **vals):
    global bad
    for k, v in vals.items():
        worst[tag + "." + k] = max(worst.get(tag + "." + k, 0.0), v)
    if not ok:
        bad += 1
        print("MISMATCH %s %s %s" % (tag, ctx, {k: "%.2e" % v for k, v in vals.items()}), flush=True)


for it in range(ncase):
    g = torch.Generator(device=dev); g.manual_seed(500 + it)
    D = int(rng.choice([16, 16, 32]))
    f16 = rng.random() < 0.25
    dt = torch.float16 if f16 else torch.float32
    B, H, W = int(rng.integers(1, 4)), int(rng.integers(34, 140)), int(rng.integers(40, 180)) // 4 * 4
    shifts = sorted(set(int(v) for v in rng.choice([1, 2, 3, 5, 9, 11, 27], size=int(rng.integers(1, 5)))))
    offsets = pkg.multi_offset(shifts, int(rng.choice([4, 8])))[:12]
    K = len(offsets)
    e = torch.randn(B, D, H, W, device=dev, generator=g).to(dt)
    ema = torch.randn(B, D, H, W, device=dev, generator=g).to(dt) if rng.random() < 0.4 else None
    cell = int(rng.integers(4, 24))
    lab = torch.randint(0, 6, (B, (H + cell - 1) // cell, (W + cell - 1) // cell), device=dev, generator=g)
    lab = lab.repeat_interleave(cell, 1).repeat_interleave(cell, 2)[:, :H, :W].contiguous().to(torch.int32)
    ctx = "case %d D=%d f16=%d B=%d %dx%d shifts=%s K=%d ema=%d" % (it, D, f16, B, H, W, shifts, K, ema is not None)
    tol_g = 5e-3 if f16 else 1e-4  # f16: the gradient itself is stored in half precision (two roundings may differ by an ulp)

    # ---- (a) labels-in against targets + tensor path
    t, m, w = pkg.gen_targets(lab, offsets, padding=True)
    res = []
    for path in ("tensor", "labels"):
        et = e.clone().requires_grad_(True)
        if path == "tensor":
            out = pkg.ema_embedding_loss(et, ema, t, w, m, crit, offsets, affs0_weight=2) if ema is not None else \
                pkg.embedding_loss(et, t, w, m, crit, offsets)
        else:
            out = pkg.ema_embedding_loss_from_labels(et, ema, lab, crit, offsets, affs0_weight=2) if ema is not None else \
                pkg.embedding_loss_from_labels(et, lab, crit, offsets)
        (out[0] * 1.25).backward()
        res.append((out[0].item(), out[1], et.grad))
    da = float((res[0][1] - res[1][1]).abs().max())
    dl = abs(res[0][0] - res[1][0]) / max(abs(res[0][0]), 1e-9)
    dg = rel(res[1][2], res[0][2])
    note("labels", da < 1e-5 and dl < 1e-5 and dg < tol_g, ctx, affs=da, loss=dl, grad=dg)

    # ---- (b) cross kernels (where the dispatch takes them: f32, axis-aligned, wide enough) against the tiled kernels
    res = []
    for xdma in ("0", "1"):
        pkg._lib.set_switch("PEA_FWD_XDMA", xdma); pkg._lib.set_switch("PEA_BWD_XDMA", xdma)
        et = e.clone().requires_grad_(True)
        out = pkg.ema_embedding_loss(et, ema, t, w, m, crit, offsets) if ema is not None else pkg.embedding_loss(et, t, w, m, crit, offsets)
        (out[0] * 0.5).backward()
        res.append((out[0].item(), out[1], et.grad))
    pkg._lib.set_switch("PEA_FWD_XDMA", None); pkg._lib.set_switch("PEA_BWD_XDMA", None)
    da = float((res[0][1] - res[1][1]).abs().max())
    dl = abs(res[0][0] - res[1][0]) / max(abs(res[0][0]), 1e-9)
    dg = rel(res[1][2], res[0][2])
    note("cross", da < 1e-5 and dl < 1e-5 and dg < tol_g, ctx, affs=da, loss=dl, grad=dg)

    # ---- (c) head against torch's convolution
    D_h = int(rng.choice([16, 32]))
    C = int(rng.choice([28, 32, 36, 48, 64, 80, 128, 256] if D_h == 16 else [32, 64, 128, 256]))
    sp = (int(rng.integers(1, 4)), int(rng.integers(3, 40)), int(rng.integers(3, 50))) if rng.random() < 0.3 else \
        (int(rng.integers(3, 90)), int(rng.integers(3, 120)))
    Bh = int(rng.integers(1, 4))
    x = torch.randn((Bh, C) + sp, device=dev, generator=g)
    wt = torch.randn((D_h, C) + (1,) * len(sp), device=dev, generator=g) * 0.2
    bs = torch.randn(D_h, device=dev, generator=g)
    up = torch.randn((Bh, D_h) + sp, device=dev, generator=g)
    conv = F.conv3d if len(sp) == 3 else F.conv2d
    outs = []
    for mine in (False, True):
        xr, wr, br = x.clone().requires_grad_(True), wt.clone().requires_grad_(True), bs.clone().requires_grad_(True)
        eh = pkg.EmbeddingHead.apply(xr, wr, br) if mine else conv(xr, wr, br)
        (eh * up).sum().backward()
        outs.append((eh.detach(), xr.grad, wr.grad, br.grad))
    d = [rel(outs[1][k], outs[0][k]) for k in range(4)]
    note("head", max(d) < 5e-5, "case %d C=%d D=%d B=%d sp=%s" % (it, C, D_h, Bh, sp), e=d[0], dx=d[1], dW=d[2], db=d[3])

    # ---- (d) the loss section: one autograd node (two-phase dual kernels, second stream) and the labels-in section
    #          against the call-by-call composition
    if it % 3 == 0:
        nb_half = 2
        offs_s = pkg.multi_offset([1, 3, 5, 9, 27][:int(rng.integers(3, 6))], 4)
        Hs, Ws = int(rng.integers(3, 10)) * 16, int(rng.integers(3, 12)) * 16
        Bs = int(rng.integers(1, 3))
        Ds = int(rng.choice([16, 32]))
        es = torch.randn(Bs, Ds, Hs, Ws, device=dev, generator=g)
        emas = torch.randn(Bs, Ds, Hs, Ws, device=dev, generator=g)
        labs = torch.randint(0, 5, (Bs, Hs // 8 + 1, Ws // 8 + 1), device=dev, generator=g).repeat_interleave(8, 1).repeat_interleave(8, 2)
        labs = labs[:, :Hs, :Ws].contiguous().to(torch.int32)
        lab_l = [labs[:, ::2 ** j, ::2 ** j].contiguous() for j in range(5)]
        emds = [torch.randn(Bs, Ds, Hs >> (j + 1), Ws >> (j + 1), device=dev, generator=g) for j in range(4)]
        tt, mm, ww = pkg.gen_targets(lab_l[0], offs_s, padding=True)
        downs = []
        for j in range(1, 5):
            k = nb_half * (5 - j)
            if k > len(offs_s):
                break
            tj, mj, wj = pkg.gen_targets(lab_l[j], offs_s[:k], padding=True)
            downs.append(torch.cat([tj, wj, mj.float()], dim=1))
        if len(downs) == 4:
            kw = dict(affs0_weight=2, deep_weight=int(rng.integers(0, 3)), self_emb=0.8, cross_emb=1.2)
            rs = []
            for which in ("composed", "node", "labels"):
                xs = [es.clone().requires_grad_(True)] + [v.clone().requires_grad_(True) for v in emds]
                if which == "labels":
                    loss, pred, _ = pkg.cvppp_loss_section_from_labels(xs[0], xs[1:], emas, lab_l[0], lab_l[1:], crit, offs_s, nb_half, **kw)
                else:
                    fn = pkg.cvppp_loss_section_composed if which == "composed" else pkg.cvppp_loss_section
                    loss, pred, _ = fn(xs[0], xs[1:], emas, tt, ww, mm, downs, crit, offs_s, nb_half, **kw)
                (loss * 0.5).backward()
                rs.append((loss.item(), pred, [v.grad for v in xs]))
            for k_, nm in ((1, "section_node"), (2, "section_labels")):
                dl = abs(rs[k_][0] - rs[0][0]) / abs(rs[0][0])
                da = float((rs[k_][1] - rs[0][1]).abs().max())
                dg = max(rel(a_, b_) for a_, b_ in zip(rs[k_][2], rs[0][2]))
                note(nm, dl < 1e-5 and da < 1e-5 and dg < 1e-4, "case %d D=%d B=%d %dx%d K=%d %s" % (it, Ds, Bs, Hs, Ws, len(offs_s), kw), loss=dl, affs=da, grad=dg)

print("fuzz_paths: %d cases, %d mismatches; worst %s" % (ncase, bad, {k: "%.1e" % v for k, v in sorted(worst.items())}))
sys.exit(1 if bad else 0)
