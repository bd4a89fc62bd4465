#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own functions.

Run in the build container only (needs /root/reference; the GPU box has neither):
    python tests/golden/make_golden.py

What is imported from the reference (nothing is copied; the files are loaded where they lie):
    scripts_cvppp/loss/loss_embedding_mse.py   embedding_loss, ema_embedding_loss, embedding2affs
    scripts_ac3ac4/loss/loss_embedding_mse.py  embedding_loss_norm1/5, ema_..., inf_...
    scripts_cvppp/loss/loss.py                 WeightedMSE  (run unmodified; its `.cuda()` call at
                                               loss.py:116 is made a no-op on this GPU-less host)
    scripts_cvppp/utils/affinity_ours.py       multi_offset, gen_affs_ours
    scripts_cvppp/model/unet2d_residual.py     OutConv (the 2D embedding head)
    scripts_ac3ac4/model/basic.py              conv3dBlock (the 3D embedding head, 1x1x1)
    scripts_cvppp/data/data_segmentation.py    not importable here (needs skimage): weight_binary_ratio
                                               (:205-228) is restated below for realistic class-balance weights.

Every fixture is data only: seeded inputs, and the reference's outputs (loss, per-offset losses,
affinity maps, autograd gradients).  Inputs are stored, not re-derived, so the fixtures do not
depend on any RNG implementation.
"""
import importlib.util
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def load(name, path):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, path))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


ref2d = load("ref_mse2d", "scripts_cvppp/loss/loss_embedding_mse.py")
ref3d = load("ref_mse3d", "scripts_ac3ac4/loss/loss_embedding_mse.py")
refloss = load("ref_loss", "scripts_cvppp/loss/loss.py")
refaff = load("ref_aff", "scripts_cvppp/utils/affinity_ours.py")

# WeightedMSE.weighted_mse_loss calls norm_term.cuda() (loss.py:116); on a CPU-only host make that
# call the identity so the reference class itself runs, arithmetic untouched.
torch.Tensor.cuda = lambda self, *a, **k: self
criterion = refloss.WeightedMSE()


def weight_binary_ratio(label, alpha=1.0):
    """data_segmentation.py:205-228 (mask=None branch)."""
    if label.max() == label.min():
        return np.ones_like(label, np.float32)
    label = (label != 0).astype(int)
    f = float(label.sum()) / np.prod(label.shape)
    f = np.clip(f, 5e-2, 0.99)
    if f > 0.5:
        w = label + alpha * f / (1 - f) * (1 - label)
    else:
        w = alpha * (1 - f) / f * label + (1 - label)
    return w.astype(np.float32)


def blocky_labels(rng, H, W, cell=8, n=12):
    """Random instance map: background 0 plus n instances, piecewise constant on a coarse grid."""
    gh, gw = -(-H // cell), -(-W // cell)
    coarse = rng.integers(0, n + 1, size=(gh, gw))
    lab = np.kron(coarse, np.ones((cell, cell), dtype=coarse.dtype))[:H, :W]
    return lab.astype(np.float32)  # gen_affs_ours subtracts labels; the reference feeds float/uint arrays


def targets_2d(rng, B, H, W, offsets):
    t = np.zeros((B, len(offsets), H, W), np.float32)
    m = np.zeros((B, len(offsets), H, W), np.uint8)
    w = np.zeros((B, len(offsets), H, W), np.float32)
    for b in range(B):
        lab = blocky_labels(rng, H, W)
        tb, mb = refaff.gen_affs_ours(lab, offsets, ignore=False, padding=True)
        t[b], m[b] = tb, mb
        for i in range(len(offsets)):
            w[b, i] = weight_binary_ratio(tb[i])
    return t, w, m


def case_targets(name, seed, B, H, W, shifts, nb):
    """label image -> what the data provider feeds the loss (scripts_cvppp/data/data_provider.py:204-225):
    gen_affs_ours(ignore=False) run by the reference itself for padding=True and padding=False, plus the
    per-channel class-balance weights (weight_binary_ratio, restated above: its module needs skimage)."""
    rng = np.random.default_rng(seed)
    offsets = refaff.multi_offset(shifts, neighbor=nb)
    labels = np.stack([blocky_labels(rng, H, W, cell=6, n=9) for _ in range(B)])
    labels[-1, : H // 3] = 0  # a large background area and, below, one uniform channel (weights all 1)
    out = {}
    for padding in (True, False):
        t = np.zeros((B, len(offsets), H, W), np.float32)
        m = np.zeros((B, len(offsets), H, W), np.uint8)
        w = np.zeros((B, len(offsets), H, W), np.float32)
        for b in range(B):
            t[b], m[b] = refaff.gen_affs_ours(labels[b], offsets, ignore=False, padding=padding)
            for i in range(len(offsets)):
                w[b, i] = weight_binary_ratio(t[b, i])
        tag = "pad" if padding else "nopad"
        out["target_" + tag], out["mask_" + tag], out["weight_" + tag] = t, m, w
    save(name, labels=labels.astype(np.int32), offsets=np.asarray(offsets, np.int32), **out)


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def save(name, **kw):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in kw.items()})
    print("%-28s %7.1f KB" % (name, os.path.getsize(path) / 1024))


def case_2d(name, seed, B, D, H, W, shifts, nb, K=None, mode="ours", zero_px=False, scale=1.0):
    rng = np.random.default_rng(seed)
    offsets = refaff.multi_offset(shifts, neighbor=nb)
    if K:
        offsets = offsets[:K]
    e = (rng.standard_normal((B, D, H, W)) * scale).astype(np.float32)
    if zero_px:  # a pixel whose norm is below eps: F.normalize divides by eps
        e[0, :, 3, 4] = 0.0
        e[-1, :, H - 1, W - 1] = 1e-14
    t, w, m = targets_2d(rng, B, H, W, offsets)
    et = T(e).requires_grad_(True)
    loss, affs, all_loss = ref2d.embedding_loss(et, T(t), T(w), T(m), criterion, offsets, affs0_weight=1, mode=mode)
    loss.backward()
    inf = ref2d.embedding2affs(T(e), offsets, mode=mode)
    save(name, kind="2d_self", mode=mode, offsets=np.array(offsets, np.int32), e=e, target=t, weight=w, mask=m,
         loss=np.float32(loss.item()), all_loss=np.array(all_loss, np.float64), affs=affs.detach().numpy(),
         affs_infer=inf.numpy(), grad=et.grad.numpy())


def case_2d_ema(name, seed, B, D, H, W, shifts, nb, affs0_weight, detach):
    rng = np.random.default_rng(seed)
    offsets = refaff.multi_offset(shifts, neighbor=nb)
    e = rng.standard_normal((B, D, H, W)).astype(np.float32)
    ema = (e + 0.5 * rng.standard_normal((B, D, H, W))).astype(np.float32)
    t, w, m = targets_2d(rng, B, H, W, offsets)
    et = T(e).requires_grad_(True)
    mt = T(ema).requires_grad_(not detach)
    loss, affs = ref2d.ema_embedding_loss(et, mt, T(t), T(w), T(m), criterion, offsets, affs0_weight=affs0_weight)
    loss.backward()
    kw = dict(kind="2d_ema", offsets=np.array(offsets, np.int32), e=e, ema=ema, target=t, weight=w, mask=m,
              affs0_weight=np.float32(affs0_weight), detach=np.bool_(detach), loss=np.float32(loss.item()),
              affs=affs.detach().numpy(), grad=et.grad.numpy())
    if not detach:
        kw["grad_ema"] = mt.grad.numpy()
    save(name, **kw)


def targets_3d(rng, B, K, Z, Y, X):
    t = (rng.random((B, K, Z, Y, X)) < 0.7).astype(np.float32)
    w = np.stack([np.stack([weight_binary_ratio(t[b, i]) for i in range(K)]) for b in range(B)])
    return t, w.astype(np.float32)


def case_3d(name, seed, B, D, Z, Y, X, which, affs0_weight=1, shift=1, ema=False):
    rng = np.random.default_rng(seed)
    K = 3 if which == "norm1" else 12
    e = rng.standard_normal((B, D, Z, Y, X)).astype(np.float32)
    t, w = targets_3d(rng, B, K, Z, Y, X)
    et = T(e).requires_grad_(True)
    kw = {}
    if ema:
        em = (e + 0.5 * rng.standard_normal(e.shape)).astype(np.float32)
        fn = ref3d.ema_embedding_loss_norm1 if which == "norm1" else ref3d.ema_embedding_loss_norm5
        loss, affs = fn(et, T(em), T(t), T(w), criterion, affs0_weight=affs0_weight, shift=shift)
        kw["ema"] = em
    else:
        fn = ref3d.embedding_loss_norm1 if which == "norm1" else ref3d.embedding_loss_norm5
        loss, affs = fn(et, T(t), T(w), criterion, affs0_weight=affs0_weight, shift=shift)
        inf = ref3d.inf_embedding_loss_norm1(T(e), shift=shift) if which == "norm1" else ref3d.inf_embedding_loss_norm5(T(e))
        kw["affs_infer"] = inf.numpy()
    loss.backward()
    save(name, kind="3d_" + which + ("_ema" if ema else ""), e=e, target=t, weight=w, shift=np.int32(shift),
         affs0_weight=np.float32(affs0_weight), loss=np.float32(loss.item()), affs=affs.detach().numpy(),
         grad=et.grad.numpy(), **kw)


def case_3d_norm6(name, seed, B, D, Z, Y, X, offsets, ema=False, both=False):
    """embedding_loss_norm6 / ema_embedding_loss_norm6 (scripts_ac3ac4/loss/loss_embedding_mse.py:346-366): generic 3D
    offsets, replicate-padded shifts (shift_tensor :294-344), ONE criterion call over all channels."""
    rng = np.random.default_rng(seed)
    K = len(offsets)
    e = rng.standard_normal((B, D, Z, Y, X)).astype(np.float32)
    t, w = targets_3d(rng, B, K, Z, Y, X)
    et = T(e).requires_grad_(True)
    kw = {}
    if ema:
        em = (e + 0.5 * rng.standard_normal(e.shape)).astype(np.float32)
        mt = T(em).requires_grad_(both)
        loss, affs = ref3d.ema_embedding_loss_norm6(et, mt, T(t), T(w), criterion, shift=[list(o) for o in offsets])
        kw["ema"] = em
    else:
        loss, affs = ref3d.embedding_loss_norm6(et, T(t), T(w), criterion, shift=[list(o) for o in offsets])
    loss.backward()
    if ema and both:
        kw["grad_ema"] = mt.grad.numpy()
    save(name, kind="3d_norm6" + ("_ema" if ema else ""), e=e, target=t, weight=w, offsets=np.asarray(offsets, np.int32),
         loss=np.float32(loss.item()), affs=affs.detach().numpy(), grad=et.grad.numpy(), **kw)


def case_head(name, seed, B, C, D, spatial, bias=True):
    """The embedding head: OutConv (2D, unet2d_residual.py:67-74) or conv3dBlock([C],[D],[(1,1,1)]) (3D, basic.py:114-127;
    model_superhuman.py:437) with seeded parameters; outputs and autograd gradients of sum(e * upstream)."""
    torch.manual_seed(seed)
    if len(spatial) == 2:
        refm = load("ref_unet2d", "scripts_cvppp/model/unet2d_residual.py")
        head = refm.OutConv(C, D)
        conv = head.conv
    else:
        sys.path.insert(0, os.path.join(REF, "scripts_ac3ac4"))
        refb = load("ref_basic3d", "scripts_ac3ac4/model/basic.py")
        head = refb.conv3dBlock([C], [D], [(1, 1, 1)], bias=[bias], init_mode='kaiming_normal')
        conv = head[0]
    with torch.no_grad():  # non-trivial parameters whatever the init mode does
        conv.weight.copy_(torch.randn_like(conv.weight) * 0.3)
        if conv.bias is not None:
            conv.bias.copy_(torch.randn_like(conv.bias))
    x = torch.randn((B, C) + tuple(spatial), requires_grad=True)
    up = torch.randn((B, D) + tuple(spatial))
    e = head(x)
    (e * up).sum().backward()
    save(name, x=x.detach().numpy(), weight=conv.weight.detach().numpy().reshape(D, C),
         bias=(conv.bias.detach().numpy() if conv.bias is not None else np.zeros(0, np.float32)),
         upstream=up.numpy(), e=e.detach().numpy(), dx=x.grad.numpy(), dW=conv.weight.grad.numpy().reshape(D, C),
         db=(conv.bias.grad.numpy() if conv.bias is not None else np.zeros(0, np.float32)))


def case_model(name, seed, shape, nfeatures, emd):
    """the reference's ResidualUNet2D_deep (scripts_cvppp/model/unet2d_residual.py:279-353) with small widths: its state_dict
    (the layout checkpoints are saved in), a seeded input and the six outputs in training mode (BatchNorm on batch statistics)"""
    refmodel = load("ref_unet2d", "scripts_cvppp/model/unet2d_residual.py")
    torch.manual_seed(seed)
    net = refmodel.ResidualUNet2D_deep(in_channels=3, out_channels=2, nfeatures=nfeatures, emd=emd)
    x = torch.randn(*shape)
    outs = net(x)
    sd = {"sd/" + k: v.detach().numpy() for k, v in net.state_dict().items()}
    save(name, x=x.numpy(), nfeatures=np.array(nfeatures), emd=np.int32(emd), keys=np.array(list(net.state_dict().keys())),
         **{"out%d" % i: o.detach().numpy() for i, o in enumerate(outs)}, **sd)


def case_seg_to_aff(name, seed, Z, Y, X):
    """the reference's seg_to_aff (scripts_ac3ac4/data/data_affinity.py:53-102) as the 3D data provider calls it
    (data_provider_labeled_deep.py:233-256): the 3-edge graph with pad='replicate' (deep-supervision scales) and the twelve
    channels of the norm5 stencil with pad='' (four 3-edge calls concatenated)"""
    refaff3 = load("ref_data_affinity", "scripts_ac3ac4/data/data_affinity.py")
    rng = np.random.default_rng(seed)
    seg = np.kron(rng.integers(0, 5, size=(-(-Z // 3), -(-Y // 5), -(-X // 6))), np.ones((3, 5, 6), dtype=np.int64))[:Z, :Y, :X]
    seg[:, : Y // 4] = 0  # a background slab: the "both > 0" rule matters
    seg = seg.astype(np.uint16)
    a3 = refaff3.seg_to_aff(seg)  # pad='replicate'
    nh = lambda a, b, c: np.asarray([-a, 0, 0, 0, -b, 0, 0, 0, -c]).reshape((3, 3))
    a12 = np.concatenate([refaff3.seg_to_aff(seg, pad=''), refaff3.seg_to_aff(seg, nh(2, 3, 3), pad=''),
                          refaff3.seg_to_aff(seg, nh(3, 9, 9), pad=''), refaff3.seg_to_aff(seg, nh(4, 27, 27), pad='')], axis=0)
    save(name, seg=seg.astype(np.int32), aff3_replicate=a3, aff12_nopad=a12)


def case_stitch_weight(name, out_size, sigma=0.2, mu=0.0):
    """Provider_valid.get_weight (scripts_ac3ac4/data/provider_valid.py:306-318), num_z >= 18 branch.  The module needs cv2 / h5py
    (absent here), so the formula is restated in THIS script, outside the product: the stitcher test compares the product's
    weights with this stored array instead of with themselves."""
    zz, yy, xx = np.meshgrid(np.linspace(-1, 1, out_size[0], dtype=np.float32), np.linspace(-1, 1, out_size[1], dtype=np.float32),
                             np.linspace(-1, 1, out_size[2], dtype=np.float32), indexing='ij')
    dd = np.sqrt(zz * zz + yy * yy + xx * xx)
    weight = 1e-6 + np.exp(-((dd - mu) ** 2 / (2.0 * sigma ** 2)))
    save(name, out_size=np.array(out_size), weight=weight[np.newaxis, ...].astype(np.float32))


def case_section(name, seed, B, D, H, W):
    """the loss section of the training loop, scripts_cvppp/main.py:284-310, run by the reference's own embedding_loss /
    ema_embedding_loss in that order on a five-scale pyramid (deep_weight 1, self_emb = cross_emb = 1, affs0_weight 1, the EMA
    operand detached as convert_consistency_flip leaves it): the six losses, the total, relu(pred) and the five gradients"""
    rng = np.random.default_rng(seed)
    offsets = refaff.multi_offset([1, 3, 5, 9, 27], neighbor=4)
    nb_half = 2
    labs, embs, downs = [], [], []
    lab0 = np.stack([blocky_labels(rng, H, W, cell=8, n=7) for _ in range(B)])
    for j in range(5):
        lab = lab0[:, ::2 ** j, ::2 ** j]
        labs.append(lab.astype(np.int32))
        embs.append(rng.standard_normal((B, D, H >> j, W >> j)).astype(np.float32))
    ema = rng.standard_normal((B, D, H, W)).astype(np.float32)
    ks = [len(offsets), nb_half * 4, nb_half * 3, nb_half * 2, nb_half * 1]
    twm = []
    for j in range(5):
        offs = offsets[:ks[j]]
        t = np.zeros((B, ks[j]) + labs[j].shape[1:], np.float32); m = np.zeros_like(t, dtype=np.uint8); w = np.zeros_like(t)
        for b in range(B):
            t[b], m[b] = refaff.gen_affs_ours(labs[j][b].astype(np.float32), offs, ignore=False, padding=True)
            for i in range(ks[j]):
                w[b, i] = weight_binary_ratio(t[b, i])
        twm.append((t, w, m))
    et = [T(e).requires_grad_(True) for e in embs]
    emat = T(ema)  # detached
    losses = []
    for j in range(1, 5):  # emd1..emd4 = scales 1/2 .. 1/16 with offsets[:8], [:6], [:4], [:2]  (main.py:284-287)
        t, w, m = twm[j]
        down = torch.cat([T(t), T(w), T(m).float()], dim=1)
        k = ks[j]
        lj, _, _ = ref2d.embedding_loss(et[j], down[:, 0:k], down[:, k:2 * k], down[:, 2 * k:3 * k], criterion, offsets[:k], affs0_weight=1, mode='ours')
        losses.append(lj)
    t, w, m = twm[0]
    l0, pred, _ = ref2d.embedding_loss(et[0], T(t), T(w), T(m), criterion, offsets, affs0_weight=1, mode='ours')
    lx, _ = ref2d.ema_embedding_loss(et[0], emat, T(t), T(w), T(m), criterion, offsets, affs0_weight=1, mode='ours')
    total = (losses[0] + losses[1] + losses[2] + losses[3] + l0) * 1.0 + lx * 1.0
    total.backward()
    out = {"emb%d" % j: embs[j] for j in range(5)}
    out.update({"lab%d" % j: labs[j] for j in range(5)})
    for j in range(5):
        out["t%d" % j], out["w%d" % j], out["m%d" % j] = twm[j]
    out.update({"grad%d" % j: et[j].grad.numpy() for j in range(5)})
    save(name, ema=ema, offsets=np.array(offsets, np.int32), total=np.float32(total.item()),
         losses=np.array([l0.item()] + [l.item() for l in losses] + [lx.item()], np.float64),  # self@1, emd1..emd4, cross
         pred=torch.relu(pred).detach().numpy(), **out)


def case_section_3d(name, seed, B, D, Z, Y, X, mode):
    """the loss section of scripts_ac3ac4/main.py:219-237, run by the reference's own functions in that order: the full-resolution
    self + EMA cross loss (norm1 or norm5), four norm1 losses on the deep-supervision heads paired emd1<->down4 .. emd4<->down1,
    loss.backward(), then the border fill of the three shift-1 channels and relu on pred"""
    rng = np.random.default_rng(seed)
    K = 3 if mode == 1 else 12
    dims = [(Z, Y, X)] + [(max(Z >> 0, 1), Y >> j, X >> j) for j in range(1, 5)]   # the heads keep z, halve y / x (superhuman U-Net)
    emb = rng.standard_normal((B, D) + dims[0]).astype(np.float32)
    ema = rng.standard_normal((B, D) + dims[0]).astype(np.float32)
    target = (rng.random((B, K) + dims[0]) < 0.7).astype(np.float32)
    weight = (0.5 + rng.random((B, K) + dims[0])).astype(np.float32)
    # emd1 .. emd4: coarsest first (model_superhuman.py returns them in that order); downN = [target(3) | weight(3)] at scale N
    emds = [rng.standard_normal((B, D) + dims[4 - j]).astype(np.float32) for j in range(4)]
    downs = [np.concatenate([(rng.random((B, 3) + dims[j]) < 0.7).astype(np.float32), (0.5 + rng.random((B, 3) + dims[j])).astype(np.float32)], axis=1)
             for j in range(1, 5)]     # down1 .. down4
    et, emt = T(emb).requires_grad_(True), [T(e).requires_grad_(True) for e in emds]
    f_self = ref3d.embedding_loss_norm1 if mode == 1 else ref3d.embedding_loss_norm5
    f_ema = ref3d.ema_embedding_loss_norm1 if mode == 1 else ref3d.ema_embedding_loss_norm5
    l0, pred = f_self(et, T(target), T(weight), criterion, affs0_weight=1)
    lx, _ = f_ema(et, T(ema), T(target), T(weight), criterion, affs0_weight=1)
    ls = []
    for emd, down in zip(emt, [T(d) for d in downs[::-1]]):   # emd1 <-> down4, ..., emd4 <-> down1 (main.py:225-228)
        l, _ = ref3d.embedding_loss_norm1(emd, down[:, :3], down[:, 3:], criterion, affs0_weight=1)
        ls.append(l)
    loss = l0 + lx + ls[0] + ls[1] + ls[2] + ls[3]
    loss.backward()
    pred = pred.clone()
    shift = 1
    pred[:, 1, :, :shift, :] = pred[:, 1, :, shift:shift * 2, :]
    pred[:, 2, :, :, :shift] = pred[:, 2, :, :, shift:shift * 2]
    pred[:, 0, :shift, :, :] = pred[:, 0, shift:shift * 2, :, :]
    pred = torch.relu(pred)
    out = {"emd%d" % (j + 1): emds[j] for j in range(4)}
    out.update({"down%d" % (j + 1): downs[j] for j in range(4)})
    out.update({"grad_emd%d" % (j + 1): emt[j].grad.numpy() for j in range(4)})
    save(name, emb=emb, ema=ema, target=target, weight=weight, mode=np.int32(mode), total=np.float32(loss.item()),
         grad_emb=et.grad.numpy(), pred=pred.detach().numpy(), **out)


def case_activation(name, seed, B, D, H, W, shifts):
    """the affinity maps of the reference's experimental loss (scripts_cvppp/loss/loss_embedding.py:33-46): CosineSimilarity
    (eps 1e-6) -> (a + 1) / 2 -> clamp; and the shipped hand-off statements on the shipped map: relu (inference.py:193), 1 - relu (seg_mutex.py:5)"""
    refexp = load("ref_loss_embedding", "scripts_cvppp/loss/loss_embedding.py")
    rng = np.random.default_rng(seed)
    offsets = refaff.multi_offset(shifts, neighbor=4)
    e = (rng.standard_normal((B, D, H, W)) * 2.0).astype(np.float32)
    half_clamp = refexp.embedding2affs(T(e), offsets).numpy()
    ours = ref2d.embedding2affs(T(e), offsets, mode='ours')
    relu = torch.relu(ours)
    save(name, e=e, offsets=np.array(offsets, np.int32), half_clamp_cos=half_clamp, relu_ours=relu.numpy(), mutex_ours=(1.0 - relu).numpy())


def case_flip(name, seed):
    """convert_consistency_flip (scripts_cvppp/data/data_consistency.py:34-45): the per-sample inverse of the EMA branch's flips /
    transpose; all eight rule combinations on a square map with distinct values"""
    refflip = load("ref_consistency", "scripts_cvppp/data/data_consistency.py")
    rng = np.random.default_rng(seed)
    rules = np.array([[a, b, c] for a in (0, 1) for b in (0, 1) for c in (0, 1)], np.float32)
    gt = rng.standard_normal((8, 3, 6, 6)).astype(np.float32)
    out = refflip.convert_consistency_flip(T(gt), T(rules)).numpy()
    save(name, gt=gt, rules=rules, out=out)


def case_full_summary(name, seed, B, D, H, W):
    """One full-size CVPPP case (B x 16 x 544 x 544, K=10): too big to store, so inputs are a closed-form
    function of the index (no RNG) and only summary statistics + samples of the outputs are kept."""
    offsets = refaff.multi_offset([1, 3, 5, 9, 27], neighbor=4)
    K = len(offsets)
    e, t, w, m = synth_full(B, D, H, W, K, seed)
    et = T(e).requires_grad_(True)
    loss, affs, all_loss = ref2d.embedding_loss(et, T(t), T(w), T(m), criterion, offsets)
    loss.backward()
    affs = affs.detach().numpy()
    grad = et.grad.numpy()
    idx = np.random.default_rng(0).integers(0, affs.size, 4096)
    gidx = np.random.default_rng(1).integers(0, grad.size, 4096)
    save(name, kind="2d_full_summary", seed=np.int64(seed), shape=np.array([B, D, H, W]), offsets=np.array(offsets, np.int32),
         loss=np.float32(loss.item()), all_loss=np.array(all_loss, np.float64),
         affs_sum=np.float64(affs.astype(np.float64).sum()), affs_sq=np.float64((affs.astype(np.float64) ** 2).sum()),
         grad_sum=np.float64(grad.astype(np.float64).sum()), grad_sq=np.float64((grad.astype(np.float64) ** 2).sum()),
         affs_idx=idx, affs_val=affs.reshape(-1)[idx], grad_idx=gidx, grad_val=grad.reshape(-1)[gidx])


def case_3d_summary(name, seed, B, D, Z, Y, X, which, affs0_weight=1):
    """A 3D case sized for the LDS-DMA kernels that march along z (csrc/pea_zmarch.h needs Y >= 43, X >= 96, X % 4 == 0: 4-6 MB of
    tensors): inputs are a closed-form function of the index (utils/synth.py, no RNG), only summary statistics and 4096 sampled
    values of the reference's outputs are kept."""
    sys.path.insert(0, os.path.join(OUT, "..", ".."))
    import importlib
    import __graft_entry__ as ge
    ge.load_package()
    synth = importlib.import_module(ge.PKG_NAME + ".utils.synth")
    shifts = [1, 1, 1] if which == "norm1" else [1, 1, 1, 2, 3, 3, 3, 9, 9, 4, 27, 27]
    offs = [[-s if i % 3 == a else 0 for a in range(3)] for i, s in enumerate(shifts)]
    e, t, w = synth.synth_inputs_3d(B, D, Z, Y, X, offs, seed)
    et = T(e).requires_grad_(True)
    fn = ref3d.embedding_loss_norm1 if which == "norm1" else ref3d.embedding_loss_norm5
    loss, affs = fn(et, T(t), T(w), criterion, affs0_weight=affs0_weight)
    loss.backward()
    affs, grad = affs.detach().numpy(), et.grad.numpy()
    idx = np.random.default_rng(0).integers(0, affs.size, 4096)
    gidx = np.random.default_rng(1).integers(0, grad.size, 4096)
    save(name, kind="3d_" + which + "_summary", seed=np.int64(seed), shape=np.array([B, D, Z, Y, X]), affs0_weight=np.float32(affs0_weight),
         loss=np.float32(loss.item()), affs_sum=np.float64(affs.astype(np.float64).sum()),
         affs_sq=np.float64((affs.astype(np.float64) ** 2).sum()), grad_sq=np.float64((grad.astype(np.float64) ** 2).sum()),
         affs_idx=idx, affs_val=affs.reshape(-1)[idx], grad_idx=gidx, grad_val=grad.reshape(-1)[gidx])


def synth_full(B, D, H, W, K, seed):
    """Deterministic, RNG-free synthetic inputs (pixel-embedded-affinity_amd/utils/synth.py)."""
    sys.path.insert(0, os.path.join(OUT, "..", ".."))
    import importlib
    import __graft_entry__ as ge
    ge.load_package()
    synth = importlib.import_module(ge.PKG_NAME + ".utils.synth")
    return synth.synth_inputs_2d(B, D, H, W, refaff.multi_offset([1, 3, 5, 9, 27], neighbor=4)[:K], seed)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "targets":  # only the label -> target / mask / weight fixtures
        case_targets("gtgt_2d_nb4", 41, B=2, H=45, W=52, shifts=[1, 3, 5, 9, 27], nb=4)
        case_targets("gtgt_2d_nb8", 42, B=1, H=31, W=40, shifts=[1, 3, 9], nb=8)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "norm6":  # only the replicate-border fixtures
        O6 = [[-1, 0, 0], [0, -1, 0], [0, 0, -1], [-2, 0, 0], [0, -3, 0], [0, 0, -3], [0, -3, 3], [1, 2, -2], [0, 9, 0]]
        case_3d_norm6("g3r_norm6", 51, B=2, D=16, Z=5, Y=14, X=17, offsets=O6)
        case_3d_norm6("g3r_norm6_ema", 52, B=1, D=16, Z=4, Y=12, X=13, offsets=O6, ema=True)
        case_3d_norm6("g3r_norm6_ema_both", 53, B=1, D=8, Z=4, Y=11, X=12, offsets=O6[:6], ema=True, both=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "head":  # only the embedding-head fixtures
        case_head("ghead_2d_c32_d16", 61, B=2, C=32, D=16, spatial=(19, 23))
        case_head("ghead_2d_c64_d32", 62, B=1, C=64, D=32, spatial=(9, 31))
        case_head("ghead_3d_c28_d16", 63, B=1, C=28, D=16, spatial=(3, 10, 13))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "corners":  # fixtures that pin what round 1 had pinned to the text only
        case_seg_to_aff("gseg2aff_3d", 91, Z=9, Y=40, X=42)
        case_stitch_weight("gstitch_weight_18x160x160", (18, 160, 160))
        case_section("gsection_cvppp", 92, B=2, D=16, H=48, W=64)
        case_activation("gact_2d", 93, B=1, D=16, H=40, W=72, shifts=[1, 3, 9])
        case_flip("gflip_rules", 94)
        case_section_3d("gsection_ac3ac4_norm5", 95, B=1, D=16, Z=5, Y=32, X=48, mode=5)
        case_section_3d("gsection_ac3ac4_norm1", 96, B=2, D=16, Z=4, Y=32, X=48, mode=1)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "model":  # only the backbone layout fixtures
        case_model("gmodel_resunet2d", 81, (2, 3, 48, 64), [4, 8, 12, 16, 20], 16)
        case_model("gmodel_resunet2d_odd", 82, (1, 3, 40, 40), [4, 6, 8, 10, 12], 16)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "march":  # only the fixtures sized for the z-march kernels (pea_zmarch.h)
        torch.manual_seed(0)
        case_3d_summary("g3d_norm5_march", 101, B=1, D=16, Z=9, Y=48, X=96, which="norm5", affs0_weight=2)
        case_3d_summary("g3d_norm5_march_b2", 102, B=2, D=16, Z=6, Y=43, X=100, which="norm5", affs0_weight=1)
        case_3d_summary("g3d_norm1_march", 103, B=2, D=16, Z=7, Y=40, X=72, which="norm1", affs0_weight=0.5)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "cross":  # only the fixtures sized for the LDS-DMA cross kernels (pea_xdma.h)
        torch.manual_seed(0)
        case_2d("g2d_x_k10", 71, B=1, D=16, H=50, W=100, shifts=[1, 3, 5, 9, 27], nb=4, zero_px=True)
        case_2d("g2d_x_k8", 72, B=2, D=16, H=37, W=72, shifts=[1, 3, 5, 9, 11], nb=4, K=8)
        sys.exit(0)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    # 2D, shipped CVPPP stencil (shifts 1,3,5,9,27 x neighbor 4 -> K=10), ragged sizes
    case_2d("g2d_cvppp_k10", 1, B=2, D=16, H=40, W=56, shifts=[1, 3, 5, 9, 27], nb=4)
    # neighbor=8: diagonal and mixed-sign offsets
    case_2d("g2d_nb8_k12", 2, B=1, D=16, H=48, W=37, shifts=[1, 3, 9], nb=8)
    # deep-supervision head: offsets[:8] (main.py:284), odd sizes, zero-norm pixels
    case_2d("g2d_k8_zero", 3, B=2, D=16, H=33, W=35, shifts=[1, 3, 5, 9, 27], nb=4, K=8, zero_px=True)
    # D=32 (BASELINE config 3) and the CosineSimilarity branch (mode != 'ours')
    case_2d("g2d_d32", 4, B=1, D=32, H=32, W=64, shifts=[1, 3, 5, 9, 11], nb=4)
    case_2d("g2d_cos_mode", 5, B=1, D=16, H=24, W=40, shifts=[1, 3, 5], nb=4, mode="cos", scale=3.0)
    case_2d("g2d_d64", 6, B=1, D=64, H=16, W=64, shifts=[1, 3, 5, 9], nb=4)
    case_2d("g2d_d5_generic", 7, B=1, D=5, H=19, W=23, shifts=[1, 2], nb=8)
    # shapes that reach the LDS-DMA cross kernels (D = 16, X % 4 == 0, X >= 32 + strip): the shipped stencil with ragged
    # tiles and zero-norm pixels; the deep-supervision prefix offsets[:8] with the narrow (32-pixel) strip row
    case_2d("g2d_x_k10", 71, B=1, D=16, H=50, W=100, shifts=[1, 3, 5, 9, 27], nb=4, zero_px=True)
    case_2d("g2d_x_k8", 72, B=2, D=16, H=37, W=72, shifts=[1, 3, 5, 9, 11], nb=4, K=8)
    # EMA cross loss: detached second operand (shipped, if_ema_flip) and the non-detached variant
    case_2d_ema("g2d_ema_detach", 11, B=2, D=16, H=40, W=56, shifts=[1, 3, 5, 9, 27], nb=4, affs0_weight=2, detach=True)
    case_2d_ema("g2d_ema_both", 12, B=1, D=16, H=32, W=40, shifts=[1, 3, 9], nb=8, affs0_weight=1, detach=False)
    # 3D
    case_3d("g3d_norm1", 21, B=2, D=16, Z=6, Y=20, X=24, which="norm1", affs0_weight=1)
    case_3d("g3d_norm1_s2_w", 22, B=1, D=16, Z=5, Y=12, X=17, which="norm1", affs0_weight=0.5, shift=2)
    case_3d("g3d_norm5", 23, B=1, D=16, Z=6, Y=30, X=31, which="norm5", affs0_weight=1)
    case_3d("g3d_norm5_w", 24, B=2, D=16, Z=5, Y=29, X=28, which="norm5", affs0_weight=2)
    case_3d("g3d_norm1_ema", 25, B=1, D=16, Z=6, Y=20, X=24, which="norm1", ema=True)
    case_3d("g3d_norm5_ema", 26, B=1, D=16, Z=6, Y=30, X=31, which="norm5", affs0_weight=2, ema=True)
    case_targets("gtgt_2d_nb4", 41, B=2, H=45, W=52, shifts=[1, 3, 5, 9, 27], nb=4)
    case_targets("gtgt_2d_nb8", 42, B=1, H=31, W=40, shifts=[1, 3, 9], nb=8)
    O6 = [[-1, 0, 0], [0, -1, 0], [0, 0, -1], [-2, 0, 0], [0, -3, 0], [0, 0, -3], [0, -3, 3], [1, 2, -2], [0, 9, 0]]
    case_3d_norm6("g3r_norm6", 51, B=2, D=16, Z=5, Y=14, X=17, offsets=O6)
    case_3d_norm6("g3r_norm6_ema", 52, B=1, D=16, Z=4, Y=12, X=13, offsets=O6, ema=True)
    case_3d_norm6("g3r_norm6_ema_both", 53, B=1, D=8, Z=4, Y=11, X=12, offsets=O6[:6], ema=True, both=True)
    # full CVPPP size, summary only
    case_full_summary("g2d_full544_summary", 555, B=2, D=16, H=544, W=544)
    case_head("ghead_2d_c32_d16", 61, B=2, C=32, D=16, spatial=(19, 23))
    case_head("ghead_2d_c64_d32", 62, B=1, C=64, D=32, spatial=(9, 31))
    case_head("ghead_3d_c28_d16", 63, B=1, C=28, D=16, spatial=(3, 10, 13))
    case_model("gmodel_resunet2d", 81, (2, 3, 48, 64), [4, 8, 12, 16, 20], 16)
    case_model("gmodel_resunet2d_odd", 82, (1, 3, 40, 40), [4, 6, 8, 10, 12], 16)
    case_seg_to_aff("gseg2aff_3d", 91, Z=9, Y=40, X=42)
    case_stitch_weight("gstitch_weight_18x160x160", (18, 160, 160))
    case_section("gsection_cvppp", 92, B=2, D=16, H=48, W=64)
    case_activation("gact_2d", 93, B=1, D=16, H=40, W=72, shifts=[1, 3, 9])
    case_flip("gflip_rules", 94)
    case_section_3d("gsection_ac3ac4_norm5", 95, B=1, D=16, Z=5, Y=32, X=48, mode=5)
    case_section_3d("gsection_ac3ac4_norm1", 96, B=2, D=16, Z=4, Y=32, X=48, mode=1)
    case_3d_summary("g3d_norm5_march", 101, B=1, D=16, Z=9, Y=48, X=96, which="norm5", affs0_weight=2)
    case_3d_summary("g3d_norm5_march_b2", 102, B=2, D=16, Z=6, Y=43, X=100, which="norm5", affs0_weight=1)
    case_3d_summary("g3d_norm1_march", 103, B=2, D=16, Z=7, Y=40, X=72, which="norm1", affs0_weight=0.5)
