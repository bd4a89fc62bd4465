#!/usr/bin/env python3
"""bench.py -- the hot path's headline benchmark (BASELINE.json: affinity-map Mpixels/s, fwd+bwd).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch: embedding_loss forward (affinity maps + fused
weighted-MSE loss) and its backward (d loss / d embedding), through the reference-named Python API,
i.e. through the C ABI of include/pea.h.  Workload at every N = BASELINE.json configs[1]:
CVPPP-shaped B=8 x D=16 x 544 x 544 embeddings per GPU, the shipped K=10 stencil
(shifts 1,3,5,9,27 x neighbor 4), fp32, synthetic inputs already resident in HBM.  Images are
independent units: ranks shard the batch with no data-path collective (weak scaling, 8 images per GPU).

Rank 0 prints ONE JSON line; besides the contract fields it carries
  "roofline":     the dominant kernel's algorithmic bytes / its HIP-event duration vs 8 TB/s HBM
  "cpu_baseline": the reference's op sequence (torch CPU, all host cores) on a bounded sample
"""
import argparse
import ctypes
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch

import __graft_entry__ as ge

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
SHIFTS, NEIGHBOR = [1, 3, 5, 9, 27], 4
B_PER_GPU, D, H, W = 8, 16, 544, 544

# --config: the headline workload (c2 = BASELINE.json configs[1], the default and the only one the driver runs) and the other
# single-GPU shapes of BASELINE.json / SURVEY.md section 8d, each through the same step / roofline / cpu_baseline code
CONFIGS = {
    "c1": dict(what="BASELINE configs[0]: CVPPP A1 single image, embedding_loss fwd+bwd, the shipped K=10 stencil", ndim=2, B=1, D=16, dims=(544, 544),
               shifts=[1, 3, 5, 9, 27], K=10, f16=False),
    "c1k8": dict(what="BASELINE configs[0]: CVPPP A1 single image, embedding_loss fwd+bwd, 8 affinity offsets (offsets[:8])", ndim=2, B=1, D=16,
                 dims=(544, 544), shifts=[1, 3, 5, 9, 27], K=8, f16=False),
    "c2": dict(what="BASELINE configs[1]: CVPPP A1 embedding_loss fwd+bwd", ndim=2, B=8, D=16, dims=(544, 544), shifts=[1, 3, 5, 9, 27], K=10, f16=False),
    "c3": dict(what="BASELINE configs[2]: BBBC039V1 embedding_loss fwd+bwd (per-GPU share of B=32 over 4 GPUs)", ndim=2, B=8, D=32, dims=(704, 704),
               shifts=[1, 3, 5, 9, 11], K=10, f16=False),
    "c4": dict(what="BASELINE configs[3]: AC3/AC4 embedding_loss_norm5 fwd+bwd, one 24x1024x1024 sub-volume per GPU, the reference's 12 axis offsets",
               ndim=3, B=1, D=16, dims=(24, 1024, 1024), stencil="norm5", K=12, f16=False),
    # the shape the reference TRAINS on (scripts_ac3ac4/config/ac3ac4.yaml:52 batch_size 2; data_provider_labeled_deep.py:53 crops of 18 x 160 x 160)
    "c4crop": dict(what="BASELINE configs[3], the reference's training crops: AC3/AC4 embedding_loss_norm5 fwd+bwd on 18x160x160 crops, batch 2",
                   ndim=3, B=2, D=16, dims=(18, 160, 160), stencil="norm5", K=12, f16=False),
    "c4n26": dict(what="BASELINE configs[3]: AC3/AC4 sub-volume 24x1024x1024, synthetic 26-neighbourhood (CROP_ZERO, cropped normaliser)",
                  ndim=3, B=1, D=16, dims=(24, 1024, 1024), stencil="n26", K=26, f16=False),
    "c5": dict(what="BASELINE configs[4]: D=64 embedding_loss fwd+bwd, f16 storage / f32 accumulate, offsets[:8]", ndim=2, B=8, D=64, dims=(544, 544),
               shifts=[1, 3, 5, 9, 27], K=8, f16=True),
    "c5f32": dict(what="the shape of BASELINE configs[4] with f32 storage (comparison line: D=64 on the LDS-DMA cross kernels)", ndim=2, B=8, D=64,
                  dims=(544, 544), shifts=[1, 3, 5, 9, 27], K=8, f16=False),
    # the BBBC039V1 TRAINING shape: 256 x 256 crops, batch 8 (scripts_bbbc039v1/config/bbbc039v1.yaml:48,63)
    "c3crop": dict(what="BASELINE configs[2], training crops: BBBC039V1 embedding_loss fwd+bwd on 256x256 crops", ndim=2, B=8, D=32, dims=(256, 256),
                   shifts=[1, 3, 5, 9, 11], K=10, f16=False),
    # the EMA cross loss (ema_embedding_loss / ema_embedding_loss_norm5: the second operand detached, scripts_cvppp/main.py:293,
    # scripts_ac3ac4/main.py:224): every training step of every tree calls it beside the self loss.  Algorithmic bytes: + es * D per pass
    "c2ema": dict(what="BASELINE configs[1], the EMA cross loss: ema_embedding_loss fwd+bwd (second operand detached)", ndim=2, B=8, D=16,
                  dims=(544, 544), shifts=[1, 3, 5, 9, 27], K=10, f16=False, ema=True),
    "c3ema": dict(what="BASELINE configs[2], the EMA cross loss: ema_embedding_loss fwd+bwd (second operand detached)", ndim=2, B=8, D=32,
                  dims=(704, 704), shifts=[1, 3, 5, 9, 11], K=10, f16=False, ema=True),
    "c4ema": dict(what="BASELINE configs[3], the EMA cross loss: ema_embedding_loss_norm5 fwd+bwd (second operand detached), one 24x1024x1024 sub-volume",
                  ndim=3, B=1, D=16, dims=(24, 1024, 1024), stencil="norm5", K=12, f16=False, ema=True),
    # embedding_loss_norm6 (scripts_ac3ac4/loss/loss_embedding_mse.py:346-354): generic 3D offsets with a REPLICATE
    # border -- the reference's 17-offset table shift_func(17) (utils/shift_channels.py:25-34: diagonals, reach 27).  No shipped yaml selects
    # it; it runs on the global-memory kernels (no LDS kernel takes a clamped border), and this line says what that costs
    "c4r6": dict(what="AC3/AC4 sub-volume 24x1024x1024, embedding_loss_norm6 with the reference's shift_func(17) table (REPLICATE border, one normaliser)",
                 ndim=3, B=1, D=16, dims=(24, 1024, 1024), stencil="norm6_17", K=17, f16=False),
    "c5ema": dict(what="BASELINE configs[4], the EMA cross loss: ema_embedding_loss fwd+bwd (second operand detached), f16 storage", ndim=2, B=8, D=64,
                  dims=(544, 544), shifts=[1, 3, 5, 9, 27], K=8, f16=True, ema=True),
}


def algorithmic_bytes_per_px(D, K, es=4, mask=True, ema=False):
    """fwd es*D + (4+4+[1]+4)K, bwd 2*es*D + (4+4+[1])K; es = bytes per embedding element (4, or 2 for f16
    storage).  2D f32 with the u8 mask: fwd 4D+13K, bwd 8D+9K, fwd+bwd 12D+22K; 3D (no mask): 12D+20K.  The EMA cross loss reads
    the second operand once more per pass: + es*D each (20D+22K)."""
    km = 1 if mask else 0
    fwd, bwd = es * D + (12 + km) * K, 2 * es * D + (8 + km) * K
    if ema:
        fwd, bwd = fwd + es * D, bwd + es * D
    return {"fwd": fwd, "bwd": bwd, "step": fwd + bwd}


def train_leg(pkg, dev, world, rank, dist, shared, fence, b=2, steps=8, warm=3):
    """BASELINE.json's second metric: train imgs/s of a full step at this N -- the ResidualUNet2D_deep backbone (plain PyTorch-ROCm,
    4.7 M parameters, model/unet2d_residual.py) forward + EMA forward + the labels-in loss section on the HIP kernels + backward +
    Adam, one rank per GPU under DistributedDataParallel over RCCL (bucketed gradient all-reduce overlapped with the backward;
    per-rank BatchNorm, as the reference's DataParallel has it).  b images of 544x544 per GPU (cvppp.yaml batch_size 2), weak scaling,
    synthetic images and instance labels.  Time = max over ranks of `steps` steps between barriers."""
    import importlib
    mod = importlib.import_module(ge.PKG_NAME + ".model.unet2d_residual")
    ts = importlib.import_module(ge.PKG_NAME + ".harness.train_step")
    synth = importlib.import_module(ge.PKG_NAME + ".utils.synth")
    torch.manual_seed(555)
    g = torch.Generator(device=dev).manual_seed(1000 + rank)
    x = torch.randn(b, 3, H, W, generator=g, device=dev)
    x_ema = x + 0.1 * torch.randn(b, 3, H, W, generator=g, device=dev)
    labels = torch.from_numpy(synth.synth_labels(b, (1, H, W), 555 + rank)[:, 0].copy()).to(dev).to(torch.int32)
    net = mod.ResidualUNet2D_deep(in_channels=3, out_channels=2, nfeatures=[16, 32, 64, 128, 256], emd=16).to(dev)
    model = net
    if dist is not None:
        # A rank that fails inside the DDP loop (out of memory, a shape the heads do not cover) would leave the others blocked in the
        # gradient all-reduce until the RCCL timeout, and the headline line would never be printed.  So every rank first runs ONE
        # step of its own, without any collective, and the ranks agree on the outcome before the reducer is built.
        ok = torch.ones(1, dtype=torch.int32, device="cpu" if shared else dev)
        err = None
        try:
            ts.CvpppTrainStep(net, ts.make_optimizer(net)).step(x, x_ema, labels)
            torch.cuda.synchronize()
        except Exception as ex:  # noqa: BLE001 -- reported, and the leg is dropped on every rank
            ok.zero_()
            err = ex
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            raise RuntimeError("train leg dropped on every rank: a rank failed its local step (%r)" % (err,))
        torch.manual_seed(555)
        net = mod.ResidualUNet2D_deep(in_channels=3, out_channels=2, nfeatures=[16, 32, 64, 128, 256], emd=16).to(dev)
        from torch.nn.parallel import DistributedDataParallel as DDP
        # the mask head takes no part in the shipped loss (mask_weight / ct_weight 0): its parameters never get a gradient
        DDP._set_params_and_buffers_to_ignore_for_model(net, [n for n, _ in net.named_parameters() if n.startswith("binary_seg.")]
                                                        + [n for n, _ in net.named_buffers() if n.startswith("binary_seg.")])
        model = DDP(net, device_ids=None if shared else [dev.index], broadcast_buffers=False, gradient_as_bucket_view=True)
    stepper = ts.CvpppTrainStep(model, ts.make_optimizer(net))
    for _ in range(warm):
        stepper.step(x, x_ema, labels)
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = stepper.step(x, x_ema, labels)
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if shared else dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    return {"train_imgs_per_s": round(world * b * steps / dt, 2), "train_step_ms": round(dt / steps * 1e3, 3),
            "train_config": {"model": "ResidualUNet2D_deep [16,32,64,128,256] emd 16 (PyTorch-ROCm) + HIP heads + labels-in loss section",
                             "images_per_gpu": b, "steps": steps, "optimizer": "Adam(amsgrad)",
                             "parallelism": ("ddp%d over RCCL" % world) if (dist is not None and not shared) else ("ddp%d over gloo (shared device)" % world if dist is not None else "single GPU"),
                             "loss": float(loss.item())}}


def other_config(args, pkg, dev, world, rank, dist, fence, cname=None, compact=False):
    """--config c3 | c4 | c4n26 | c5: same step timing, roofline (from the entry points' in-step HIP-event durations) and a
    bounded cpu_baseline; inputs are drawn on the GPU (torch.Generator, seed 555 + rank): N(0,1) embeddings,
    Bernoulli(0.6) targets, U(0.5,1.5) weights, Bernoulli(0.9) masks (2D).
    compact=True (the default run's "configs" field: every BASELINE config inside the line the driver records): the same step and
    in-step kernel timing on batches of COMPACT_STEPS steps, GPU legs only -- no autograd-seed / graph / CPU legs -- returned as
    {ms_per_step, ms_min, ms_max, kernel_ms, frac, pair_frac, traffic, workload}."""
    cname = cname or args.config
    c = CONFIGS[cname]
    steps = COMPACT_STEPS if compact else args.steps
    B, Dm, dims, K = (args.batch if (args.batch != B_PER_GPU and not compact) else c["B"]), c["D"], list(c["dims"]), c["K"]
    g = torch.Generator(device=dev).manual_seed(555 + rank)
    E = torch.randn([B, Dm] + dims, generator=g, device=dev)
    if c["f16"]:
        E = E.half()
    E.requires_grad_(True)
    ema = bool(c.get("ema"))
    E2 = None
    if ema:  # the EMA embedding: the detached second operand (convert_consistency_flip detaches it, data_consistency.py:36)
        E2 = torch.randn([B, Dm] + dims, generator=g, device=dev)
        E2 = E2.half() if c["f16"] else E2
    T = (torch.rand([B, K] + dims, generator=g, device=dev) < 0.6).float()
    Wt = torch.rand([B, K] + dims, generator=g, device=dev) + 0.5
    M = (torch.rand([B, K] + dims, generator=g, device=dev) < 0.9).to(torch.uint8) if c["ndim"] == 2 else None
    crit = pkg.WeightedMSE()
    L, op = pkg._lib.lib(), pkg.affinity_op
    if c["ndim"] == 2:
        offsets = pkg.multi_offset(c["shifts"], NEIGHBOR)[:K]
        spec = op.AffinitySpec(2, offsets, None, pkg._lib.BORDER_CIRCULAR, pkg._lib.NORM_BX)
    else:
        if c["stencil"] == "norm5":
            offsets = pkg.utils.affinity_ours.axis_offsets_3d(pkg.utils.affinity_ours.NORM5_SHIFTS)
        elif c["stencil"] == "norm6_17":  # shift_func(17), scripts_ac3ac4/utils/shift_channels.py:25-34
            offsets = [[-1, 0, 0], [0, -1, 0], [0, 0, -1], [-1, -1, -1], [-1, 1, 1], [-1, -1, 1], [-1, 1, -1], [0, -9, 0], [0, 0, -9],
                       [0, -9, -9], [0, 9, -9], [0, -9, -4], [0, -4, -9], [0, 4, -9], [0, 9, -4], [0, -27, 0], [0, 0, -27]]
        else:
            offsets = [[dz, dy, dx] for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1) if (dz, dy, dx) != (0, 0, 0)]
        if c["stencil"] == "norm6_17":
            spec = op.AffinitySpec(3, offsets, None, pkg._lib.BORDER_REPLICATE, pkg._lib.NORM_FULL)
        else:
            spec = op.AffinitySpec(3, offsets, None, pkg._lib.BORDER_CROP_ZERO, pkg._lib.NORM_CROPPED)

    def step():
        E.grad = None
        loss, affs, _ = op.FusedAffinityMSE.apply(E, E2, T, Wt, M, spec)
        pkg.backward(loss)  # loss.backward() seeded with a cached ones-scalar (no per-step fill kernel)

    settle = settle_gpu(step, 0.6 if compact else 1.5)
    for _ in range(3 if compact else max(args.warmup, 3)):
        step()
    bst = batch_stats(timed_batches(step, fence, steps, dist, dev), steps)  # median of N_BATCHES batches (see main())
    dt = bst["median"] * 1e-3 * steps
    # the same step started with plain loss.backward() -- what a drop-in caller of INTEGRATION.md section 2 executes (autograd seeds
    # the scalar's backward with a ones_like fill kernel); untimed for `value`, reported beside it
    def step_seed():
        E.grad = None
        loss, affs, _ = op.FusedAffinityMSE.apply(E, E2, T, Wt, M, spec)
        loss.backward()

    dt_seed = None if compact else wall_time_s(step_seed, fence, steps)
    npx = B * int(np.prod(dims))
    value = world * npx * steps / dt / 1e6
    if rank != 0:
        return None
    Ed = E.detach()
    desc = op.make_desc(spec, Ed)
    affs, G = torch.empty([B, K] + dims, device=dev), torch.empty([B, K] + dims, device=dev)
    lossv, dE, one = torch.empty(1 + K, device=dev), torch.empty_like(Ed), torch.ones((), device=dev)
    # the 1 / norm plane only where the cross backward takes it (as affinity_op.FusedAffinityMSE does)
    if ema:  # two planes (e, e_other) where the role-A cross kernels take the shape
        INV = torch.empty([2, B] + dims, device=dev) if L.pea_cross_supported(ctypes.byref(desc), 2) else None
    else:
        INV = torch.empty([B] + dims, device=dev) if L.pea_cross_supported(ctypes.byref(desc), 1) else None
    wsb = L.pea_workspace_bytes(ctypes.byref(desc))
    work = torch.empty(max(wsb, 4) // 4, device=dev)
    assert L.pea_workspace_init(ctypes.c_void_p(work.data_ptr()), wsb, None) == 0  # the loss-state block: prepared once
    P = lambda x: None if x is None else ctypes.c_void_p(x.data_ptr())
    cur = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)  # (read per call: the graph capture runs on its own stream)
    fwd = lambda: L.pea_affinity_fwd_ex(ctypes.byref(desc), P(Ed), P(E2), P(T), P(Wt), P(M), P(affs), P(G), P(INV), P(lossv), P(work), wsb, cur())
    # (the raw map goes along where the backward reads it: self loss always here; second operand: mode 4, D = 32 / 64)
    raw_in = P(affs) if (not ema or L.pea_cross_supported(ctypes.byref(desc), 4)) else None
    bwd = lambda: L.pea_affinity_bwd_ex2(ctypes.byref(desc), P(Ed), P(E2), P(G), P(INV), raw_in, P(one), P(dE), None, cur())
    in_step_times_ms(fwd, bwd, 3)
    kf, kb, kspread = in_step_batches_ms(fwd, bwd, max(10, min(steps, 50)), nb=5)
    ab = algorithmic_bytes_per_px(Dm, K, 2 if c["f16"] else 4, mask=M is not None, ema=ema)
    dom = "bwd" if kb >= kf else "fwd"
    achieved = ab[dom] * npx / (max(kf, kb) * 1e-3) / 1e9
    step_gbs = ab["step"] * npx / ((kf + kb) * 1e-3) / 1e9
    if compact:
        return {"workload": "%s: B=%d x D=%d x %s, K=%d" % (c["what"], B, Dm, "x".join(str(v) for v in dims), K),
                "ms_per_step": round(bst["median"], 5), "ms_min": round(bst["min"], 5), "ms_max": round(bst["max"], 5),
                "mpx_s": round(value, 1), "kernel_ms": {"fwd": round(kf, 5), "bwd": round(kb, 5)}, "dominant": dom,
                "frac": round(achieved / HBM_PEAK_GBS, 4), "pair_frac": round(step_gbs / HBM_PEAK_GBS, 4),
                "traffic": pmc_traffic(cname, dom)}
    out = {
        "metric": "affinity-map Mpixels/sec (fwd+bwd)", "value": round(value, 2), "unit": "Mpx/s",
        "n_gpus": world, "steps": steps, "warmup": args.warmup, "ms_per_step": round(dt / steps * 1e3, 5),
        "ms_min": round(bst["min"], 5), "ms_max": round(bst["max"], 5), "batches_ms": bst["all"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16 storage / f32 arithmetic" if c["f16"] else "f32",
        "data": "synthetic",
        "config": {"workload": "%s: B=%d per GPU x D=%d x %s, K=%d offsets" % (c["what"], B, Dm, "x".join(str(v) for v in dims), K),
                   "images_per_gpu": B, "embedding_dim": Dm, "dims": dims, "offsets": K, "sharding": "batch across ranks, no data-path collective"},
        "ms_per_step_autograd_seed": round(dt_seed / steps * 1e3, 5), "settle_steps": settle,
        "kernel_ms": {"fwd": round(kf, 5), "bwd": round(kb, 5)}, "kernel_ms_spread": kspread,
        # the launch floor as a number: the entry points' launches (forward + loss finish + backward) captured in a
        # HIP graph and replayed back to back -- no Python, no ctypes, no autograd between them
        "graph_replay_ms": graph_replay_ms(fwd, bwd, max(20, min(steps, 200))),
        # the same step through the PUBLIC API captured with pea.graphed (one graph launch per step: no Python / ctypes / autograd
        # bookkeeping per launch) -- what a caller of a small, static shape gets; host wall time per replay, replays queued back to back
        "graphed_api_ms": graphed_api_ms(pkg, op, E, E2, T, Wt, M, spec, max(20, min(steps, 200))),
        "cross_kernels": ({"fwd+bwd (second operand)": int(L.pea_cross_supported(ctypes.byref(desc), 2))} if ema else
                          {"fwd": int(L.pea_cross_supported(ctypes.byref(desc), 0)), "bwd": int(L.pea_cross_supported(ctypes.byref(desc), 1))}),
        "roofline": {"bound": "hbm", "kernel": "pea_affinity_" + dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(args.config, dom) if B == c["B"] else None,
                     "algorithmic_bytes_per_px": ab[dom], "px_per_launch": npx,
                     "ms": round(max(kf, kb), 5), "fwd_plus_bwd_GBs": round(step_gbs, 1), "fwd_plus_bwd_frac": round(step_gbs / HBM_PEAK_GBS, 4)},
    }
    if world == 1 and not args.no_cpu_baseline:
        # bounded sample: one image (2D) / one 24 x 256 x 256 block of the sub-volume (3D), same op sequence on the host cores
        if c["ndim"] == 2:
            out["cpu_baseline"] = cpu_baseline(offsets, Ed[:1].float().cpu(), T[:1].cpu(), Wt[:1].cpu(), M[:1].cpu(),
                                               what="the image" if B == 1 else "1 image of the batch",
                                               ema=None if E2 is None else E2[:1].float().cpu())
            out["speedup_vs_cpu"] = round(value / out["cpu_baseline"]["best_cpu_value"], 1)  # vs the faster CPU line
        elif c["stencil"] in ("n26", "norm6_17"):
            sl = (slice(0, 1), slice(None), slice(None), slice(0, 256), slice(0, 256))
            rep = c["stencil"] == "norm6_17"
            out["cpu_baseline"] = cpu_baseline_generic3d(offsets, Ed[sl].float().cpu().contiguous(), T[sl].cpu().contiguous(), Wt[sl].cpu().contiguous(),
                                                         what="a 24x256x256 block of the sub-volume with " +
                                                         ("the shift_func(17) table, replicate border (K=17)" if rep else "the 26-neighbourhood (K=26)"),
                                                         replicate=rep)
        else:
            sl = (slice(0, 1), slice(None), slice(None), slice(0, 256), slice(0, 256))
            out["cpu_baseline"] = cpu_baseline(None, Ed[sl].float().cpu().contiguous(), T[:, :12][sl].cpu().contiguous(), Wt[:, :12][sl].cpu().contiguous(),
                                               None, shifts3d=pkg.utils.affinity_ours.NORM5_SHIFTS,
                                               what=("a %dx%dx%d block of the sub-volume with the norm5 stencil (K=12)" % tuple(min(a, b) for a, b in zip(dims, (24, 256, 256))))
                                               if cname != "c4crop" else "one 18x160x160 crop of the batch with the norm5 stencil (K=12)",
                                               ema=None if E2 is None else E2[sl].float().cpu().contiguous())
    return out


COMPACT_STEPS = 20  # steps per timed batch of the default run's "configs" leg (21 batches each: n_batches)
# the default run's "configs" field: every BASELINE.json config that fits one GPU beside the headline (c2), the EMA cross loss of the
# headline shape and the reference's 3D training crops
LINE_CONFIGS = ("c1", "c3", "c4", "c5", "c2ema", "c4crop")


def best_event_us(run, calls=10, rounds=3, warm=3):
    """best of `rounds` batches of `calls` calls, HIP events on the launch stream, in microseconds per call (a one-off allocator stall
    inside a batch is not the section's cost)"""
    for _ in range(warm):
        run()
    return round(min(event_time_ms(run, calls) for _ in range(rounds)) * 1e3, 1)


def graphed_us(pkg, fn, xs, replays=20, rounds=3):
    """fn(*xs) captured once through the public API (pea.graphed: every stream of the section becomes one graph launch) and replayed back
    to back; wall clock per replay in microseconds, best of `rounds` batches.  None if the capture fails."""
    try:
        g = pkg.graphed(fn, *xs)
        best = None
        for _ in range(rounds):
            g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(replays):
                g.replay()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / replays
            best = ms if best is None else min(best, ms)
        del g
        return round(best * 1e6, 1)
    except Exception as ex:  # noqa: BLE001 -- an extra field, never the headline
        print("section capture failed: %r" % (ex,), file=sys.stderr)
        return None


def ac3ac4_section_us(pkg, dev, rank, B=2, dims=(18, 160, 160)):
    """The 3D training loop's loss section at the shape the reference trains on (scripts_ac3ac4/main.py:219-237; ac3ac4.yaml:52 batch 2,
    data_provider_labeled_deep.py:53 crops of 18 x 160 x 160; deep supervision at 1/2 .. 1/16 in y and x, :225-232): embedding_loss_norm5 +
    ema_embedding_loss_norm5 + four embedding_loss_norm1 + backward + border fill + relu.  Microseconds per section: one autograd node
    (eager / captured with pea.graphed), the call-by-call composition (what INTEGRATION.md section 2's import-line change runs), and the
    in-step kernel sum of the six losses' entry points for comparison."""
    g = torch.Generator(device=dev).manual_seed(900 + rank)
    Z, Y, X = dims
    crit = pkg.WeightedMSE()
    emb = torch.randn(B, 16, Z, Y, X, generator=g, device=dev)
    ema = torch.randn(B, 16, Z, Y, X, generator=g, device=dev)
    emds = [torch.randn(B, 16, Z, Y >> j, X >> j, generator=g, device=dev) for j in (4, 3, 2, 1)]  # emd1 (1/16) .. emd4 (1/2)
    target = (torch.rand(B, 12, Z, Y, X, generator=g, device=dev) < 0.6).float()
    weight = torch.rand(B, 12, Z, Y, X, generator=g, device=dev) + 0.5
    downs = [torch.cat([(torch.rand(B, 3, Z, Y >> j, X >> j, generator=g, device=dev) < 0.6).float(),
                        torch.rand(B, 3, Z, Y >> j, X >> j, generator=g, device=dev) + 0.5], dim=1) for j in (1, 2, 3, 4)]  # down1 .. down4

    def section(which, xs):
        """composed: call by call + finish_pred_3d_; one_node: one autograd node + finish_pred_3d_; finished: one node that hands the
        map out finished (the forward clamps it, only the border slices are touched afterwards)"""
        if which == "composed":
            loss, pred = pkg.ac3ac4_loss_section_composed(xs[0], list(xs[1:]), ema, target, weight, downs, crit, embedding_mode=5)
        else:
            loss, pred = pkg.ac3ac4_loss_section(xs[0], list(xs[1:]), ema, target, weight, downs, crit, embedding_mode=5,
                                                 finish_pred=which == "finished")
        loss.backward()
        return loss, (pred if which == "finished" else pkg.finish_pred_3d_(pred))

    def run(which):
        section(which, [emb.detach().requires_grad_(True)] + [e.detach().requires_grad_(True) for e in emds])

    out = {"shape": "B=%d x 16 x %dx%dx%d, norm5 + ema_norm5 + 4 x norm1 (1/16 .. 1/2 in y, x)" % (B, Z, Y, X)}
    for which in ("composed", "one_node", "finished"):
        out[which] = best_event_us(lambda: run(which))
    for which in ("one_node", "finished", "composed"):
        xs = [emb.detach().clone().requires_grad_(True)] + [e.detach().clone().requires_grad_(True) for e in emds]

        def fn(*xs):
            for t in xs:
                t.grad = None
            return section(which, xs) + tuple(t.grad for t in xs)
        out[which + "_graphed"] = graphed_us(pkg, fn, xs)
    # algorithmic bytes of the six losses (12D + 20K per voxel; the cross loss + 8D) at 8 TB/s
    vox = B * Z * Y * X
    small = sum(B * Z * (Y >> j) * (X >> j) for j in (1, 2, 3, 4))
    abytes = vox * (12 * 16 + 20 * 12) + vox * (20 * 16 + 20 * 12) + small * (12 * 16 + 20 * 3)
    out["algorithmic_MB"] = round(abytes / 1e6, 1)
    out["roofline_us"] = round(abytes / (HBM_PEAK_GBS * 1e9) * 1e6, 1)
    return out


def settle_gpu(step, budget_s=1.5):
    """untimed, before the warm-up steps: a GPU that has just been handed over idles at its lowest clocks and the first few hundred
    launches also pay the allocator's and the library's first-touch costs; a 20-step timing started cold reads 25-30 % slow.  Run
    batches of 20 steps until two consecutive batches agree within 2 % (at most budget_s).  Returns the steps spent."""
    settle, prev, t_start = 0, None, time.perf_counter()
    while time.perf_counter() - t_start < budget_s:
        ts = time.perf_counter()
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        cur = time.perf_counter() - ts
        settle += 20
        if prev is not None and abs(cur - prev) <= 0.02 * prev:
            break
        prev = cur
    return settle


def gpu_state_under_load(step, dev, seconds=0.8):
    """clocks / power / temperatures of this rank's GPU WHILE the step runs, read from sysfs (hwmon of the card with the device's PCI
    address; no child process: nothing may exec behind an initialised GPU, least of all under rocprofv3).  Diagnostic only: the step's
    time comes in two states on some boxes -- DESIGN.md section 5 item 2 -- and this says what the card reported in the run.  None if
    the files are not there."""
    import glob
    try:
        pr = torch.cuda.get_device_properties(dev)
        addr = "%04x:%02x:%02x." % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        hw = None
        for card in glob.glob("/sys/class/drm/card*/device"):
            if os.path.basename(os.path.realpath(card)).startswith(addr):
                hs = glob.glob(os.path.join(card, "hwmon", "hwmon*"))
                hw = hs[0] if hs else None
                break
        if hw is None:
            return None

        def rd(name):
            try:
                return float(open(os.path.join(hw, name)).read().strip())
            except (OSError, ValueError):
                return None

        samples, t0 = [], time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            for _ in range(20):
                step()
            samples.append((rd("freq1_input"), rd("freq2_input"), rd("power1_input"), rd("temp2_input"), rd("temp3_input")))
            torch.cuda.synchronize()
        med = lambda k, sc: (lambda v: round(sorted(v)[len(v) // 2] * sc, 1) if v else None)([x[k] for x in samples if x[k] is not None])
        return {"sclk_mhz": med(0, 1e-6), "mclk_mhz": med(1, 1e-6), "power_w": med(2, 1e-6), "temp2_c": med(3, 1e-3), "temp3_c": med(4, 1e-3),
                "samples": len(samples), "source": hw}
    except Exception as ex:  # diagnostic only
        return {"error": repr(ex)[:160]}


N_BATCHES = 7  # timed batches of --steps steps each: `value` is their MEDIAN (one batch of 20 steps is 4.5 ms -- a slow
               # state of the box / process silently became the round's number; now it shows as ms_min / ms_max beside the median)


def n_batches(steps):
    """7 batches; 21 when a batch is short (--steps < 100: a 20-step batch is 4 ms, and 3 of 7 such batches were seen to take a 1 ms
    host stall each on one box -- profiles/r5_bench_s20_2.json: the median of 21 does not land on one)"""
    return N_BATCHES if steps >= 100 else 3 * N_BATCHES


def timed_batches(step, fence, steps, dist, red_dev, nb=None):
    """nb batches of EXACTLY `steps` calls of step(), each bracketed by fence() (barrier + synchronize) on both sides; per batch the MAX
    over ranks.  Returns the list of batch durations in seconds, in the order they ran.  Python's cyclic garbage collector is
    paused over the timed batches (autograd allocates per step; a generation-2 collection inside a 4 ms batch is a host stall, not
    a property of the path)."""
    import gc
    nb = nb or n_batches(steps)
    out = []
    was = gc.isenabled()  # (no gc.collect() here: 50-100 ms of host work with the GPU idle and the first batches read 5-80 % slow while its
    gc.disable()          #  clocks come back -- gpurun_out/r5_bench_s20_3.json; the collection happens before the settle phase instead)
    try:
        for _ in range(nb):
            fence()
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            fence()
            out.append(time.perf_counter() - t0)
    finally:
        if was:
            gc.enable()
    if dist is not None:
        tmax = torch.tensor(out, dtype=torch.float64, device=red_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        out = [float(v) for v in tmax.tolist()]
    return out


def batch_stats(dts, steps):
    """median / min / max of the batches, as ms per step"""
    ms = sorted(d / steps * 1e3 for d in dts)
    return {"median": ms[len(ms) // 2], "min": ms[0], "max": ms[-1], "all": [round(d / steps * 1e3, 5) for d in dts]}


def wall_time_s(step, fence, steps, warm=5):
    """host wall time of `steps` calls of step() between two fences (this rank)"""
    for _ in range(warm):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    return time.perf_counter() - t0


def graph_replay_ms(fwd, bwd, iters, per_graph=1):
    """`per_graph` x (fwd(); bwd()) (C-ABI launches on the current stream) captured once in a HIP graph; average duration of ONE
    fwd + bwd step over `iters` replays queued back to back (HIP events around the batch).  None if the capture fails."""
    try:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            fwd(); bwd()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(per_graph):
                assert fwd() == 0 and bwd() == 0
        for _ in range(5):
            g.replay()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            g.replay()
        b.record()
        b.synchronize()
        return round(a.elapsed_time(b) / iters / per_graph, 5)
    except Exception:  # noqa: BLE001 -- an extra field, never the headline
        return None


def graphed_api_ms(pkg, op, E, E2, T, Wt, M, spec, iters):
    """ms per step of `FusedAffinityMSE + pea.backward` captured once with pea.graphed (the product-level HIP-graph wrapper) and
    replayed `iters` times back to back; wall clock around the batch.  None if the capture fails."""
    try:
        def fn(E):
            E.grad = None
            loss, affs, _ = op.FusedAffinityMSE.apply(E, E2, T, Wt, M, spec)
            pkg.backward(loss)
            return loss, affs, E.grad
        g = pkg.graphed(fn, E)
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            g.replay()
        torch.cuda.synchronize()
        return round((time.perf_counter() - t0) / iters * 1e3, 5)
    except Exception:  # noqa: BLE001 -- an extra field, never the headline
        return None


def cpu_baseline_generic3d(offsets, e, t, w, what, replicate=False):
    """a 3D stencil the torch restatement has no op sequence for (the 26-neighbourhood): the oracle's C restatement with OpenMP over
    the host cores on a bounded block, CROP_ZERO border and cropped normaliser like the GPU line it sits beside"""
    orc = ge.load_oracle()
    cores = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    en, tn, wn = (np.ascontiguousarray(x.numpy()) for x in (e, t, w))
    B, Dm = en.shape[:2]
    brd, nrm = (orc.BORDER_REPLICATE, orc.NORM_FULL) if replicate else (orc.BORDER_CROP_ZERO, orc.NORM_CROPPED)
    d = orc.make_desc(B, Dm, list(en.shape[2:]), [list(o) for o in offsets], None, brd, nrm, 1e-12, 0, ndim=3)

    def once():
        t0 = time.perf_counter()
        orc.c_fwd(d, en, None, tn, wn, None)
        orc.c_bwd(d, en, None, tn, wn, None)
        return time.perf_counter() - t0

    prev = orc.c_set_threads(cores)
    once()
    dco = min(once() for _ in range(2))
    orc.c_set_threads(1)
    sub = (slice(0, 1), slice(None), slice(None), slice(0, 64), slice(0, 64))
    e1, t1, w1 = (np.ascontiguousarray(x[sub]) for x in (en, tn, wn))
    d1 = orc.make_desc(1, Dm, list(e1.shape[2:]), [list(o) for o in offsets], None, brd, nrm, 1e-12, 0, ndim=3)
    t0 = time.perf_counter()
    orc.c_fwd(d1, e1, None, t1, w1, None)
    orc.c_bwd(d1, e1, None, t1, w1, None)
    d1t = time.perf_counter() - t0
    orc.c_set_threads(prev)
    px = B * int(np.prod(en.shape[2:]))
    v = round(px / dco / 1e6, 4)
    return {"value": v, "unit": "Mpx/s", "cores": cores, "kind": "port",
            "sample": "best of 2 fwd + bwd of %s, oracle/pea_oracle.c with OpenMP (%d threads), %.2f s%s"
                      % (what, cores, dco, " -- the restatement's replicate-border BACKWARD is a serial scatter: most of this time runs on one core" if replicate else ""),
            "c_omp": {"value": v, "unit": "Mpx/s", "cores": cores, "kind": "port", "sample": "the same run"},
            "c_1thread": {"value": round(int(np.prod(e1.shape[2:])) / d1t / 1e6, 4), "unit": "Mpx/s", "cores": 1, "kind": "port",
                          "sample": "one fwd + bwd of a 24x64x64 block, oracle/pea_oracle.c, 1 thread, %.2f s" % d1t},
            "best_cpu_value": v}


def event_time_ms(fn, iters):
    """average duration of fn() over `iters` back-to-back calls, by HIP events on the launch stream"""
    s = torch.cuda.current_stream()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(s)
    for _ in range(iters):
        fn()
    b.record(s)
    b.synchronize()
    return a.elapsed_time(b) / iters


def in_step_times_ms(fwd, bwd, iters):
    """durations of the forward and of the backward INSIDE the alternating step (fwd, bwd, fwd, bwd, ...): what each kernel
    takes in the cache state the training loop leaves it in (a kernel repeated back to back finds its own inputs in the
    Infinity Cache and reads faster than it ever does in a step).  HIP events on the launch stream between the launches."""
    s = torch.cuda.current_stream()
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(iters)]
    for a, b, c in ev:
        a.record(s)
        fwd()
        b.record(s)
        bwd()
        c.record(s)
    ev[-1][2].synchronize()
    return (sum(a.elapsed_time(b) for a, b, _ in ev) / iters, sum(b.elapsed_time(c) for _, b, c in ev) / iters)


def in_step_batches_ms(fwd, bwd, iters, nb=N_BATCHES):
    """in_step_times_ms over nb batches: per entry point the median batch with the fastest and the slowest beside it"""
    rs = [in_step_times_ms(fwd, bwd, iters) for _ in range(nb)]
    f, b = sorted(r[0] for r in rs), sorted(r[1] for r in rs)
    return f[nb // 2], b[nb // 2], {"fwd_min": round(f[0], 5), "fwd_max": round(f[-1], 5), "bwd_min": round(b[0], 5), "bwd_max": round(b[-1], 5)}


# switches that select kernels (csrc/pea_k_direct.hip env_make) + the library override: with one of them set, another kernel than the
# profiled one may be running.  (Round-4 advice: the bare PEA_ prefix also caught PEA_BENCH_EXTRA, PEA_STEPS, ... and silently nulled
# the roofline's traffic field.)
KERNEL_SWITCHES = ("PEA_FORCE_DIRECT", "PEA_FWD_XDMA", "PEA_BWD_XDMA", "PEA_FWD_WG3", "PEA_BWD_PF", "PEA_BOX",
                   "PEA_H16_HW", "PEA_ZMARCH", "PEA_ZSEG", "PEA_ZM_SUP", "PEA_BOXM", "PEA_ZBLK_Y", "PEA_ZBLK_X", "PEA_BWD_REV",
                   "PEA_FWD_DUAL", "PEA_HIP_LIB")


def pmc_traffic(key, dom):
    """PMC-measured HBM bytes per launch of the dominant entry point's kernel, recorded by profiles/make_traffic.py from a
    rocprofv3 --pmc run of THESE sources (null when the kernel sources changed since)"""
    tpath = os.path.join(ge.ROOT, "profiles", "traffic.json")
    if not os.path.exists(tpath):
        return None
    tj = json.load(open(tpath))
    rec = tj.get(key, {}).get(dom)
    # (recorded with the default switches: under a PEA_* override another kernel may run)
    if any(k in os.environ for k in KERNEL_SWITCHES):
        return None
    return rec.get("bytes_per_launch") if rec and tj.get("src_sha16") == source_sha16() else None


def source_sha16():
    """hash of the kernel sources: profiles/traffic.json (PMC bytes of a profiled run) is only quoted for the code it measured"""
    h = hashlib.sha256()
    csrc = os.path.join(ge.PKG_DIR, "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".h", ".hip")):
            h.update(open(os.path.join(csrc, f), "rb").read())
    return h.hexdigest()[:16]


def isolated_time_ms(fn, iters):
    """average duration of fn() when every call is drained before the next is launched: what a profiler that separates
    dispatches reports (no overlap of one launch's tail with the next one's ramp-up)"""
    s = torch.cuda.current_stream()
    tot = 0.0
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s)
        fn()
        b.record(s)
        b.synchronize()
        tot += a.elapsed_time(b)
    return tot / iters


def cpu_baseline(offsets, e, t, w, m, budget_s=20.0, shifts3d=None, what=None, ema=None):
    """The reference's arithmetic (F.normalize -> K x roll/mul/sum -> WeightedMSE -> autograd backward; 3D: the cropped slices of
    embedding_loss_norm5) as the oracle's torch-CPU restatement, on all host cores, on a bounded sample of the workload (numpy
    arrays or CPU tensors); at most ~budget_s of CPU work.  The only place bench.py touches the oracle."""
    orc = ge.load_oracle()
    # the GPU box gives one GPU a 16-core CPU share; more threads than that only oversubscribes
    cores = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    torch.set_num_threads(cores)
    et, tt, wt = (torch.as_tensor(x) for x in (e, t, w))
    mt = None if m is None else torch.as_tensor(m)
    emt = None if ema is None else torch.as_tensor(ema)  # the EMA cross loss: the detached second operand

    def one():
        x = et.clone().requires_grad_(True)
        if shifts3d is not None:
            loss = orc.torch_embedding_loss_3d(x, tt, wt, shifts3d, ema=emt)[0]
        else:
            loss = orc.torch_embedding_loss(x, tt, wt, mt, offsets, ema=emt)[0]
        loss.backward()
        return float(loss.detach())

    t0 = time.perf_counter()
    one()  # warm-up (allocator, thread pool)
    first = time.perf_counter() - t0
    iters = int(max(1, min(5, budget_s // max(first, 1e-3) - 1)))
    t0 = time.perf_counter()
    for _ in range(iters):
        one()
    dt = (time.perf_counter() - t0) / iters
    px = et.shape[0] * int(np.prod(et.shape[2:]))
    what = what or "the same B=%d x %d x %dx%d K=%d batch" % (et.shape[0], et.shape[1], et.shape[2], et.shape[3], len(offsets))
    out = {"value": round(px / dt / 1e6, 4), "unit": "Mpx/s", "cores": cores, "kind": "port",
           "sample": "%d timed fwd+bwd iterations (after 1 warm-up) of %s, oracle/pea_oracle.py torch restatement, %.2f s/iter" % (iters, what, dt)}
    # the same arithmetic as the oracle's C restatement (oracle/pea_oracle.c: fused loops, no temporaries): one scalar thread, and
    # OpenMP over all host cores -- SURVEY.md section 8d's "second, stronger CPU line"; speedup_vs_cpu is quoted against the FASTER
    # of the torch port and this one
    en, tn, wn = (np.ascontiguousarray(x.numpy()) for x in (et, tt, wt))
    mn = None if mt is None else np.ascontiguousarray(mt.numpy())
    on = None if emt is None else np.ascontiguousarray(emt.numpy())
    d = orc.desc_3d(en, shifts3d) if shifts3d is not None else orc.desc_2d(en, offsets)

    def c_once(sl):
        t0 = time.perf_counter()
        cut = (lambda a: None if a is None else a[sl]) if sl is not None else (lambda a: a)
        orc.c_fwd(d if sl is None else d1, cut(en), cut(on), cut(tn), cut(wn), cut(mn))
        orc.c_bwd(d if sl is None else d1, cut(en), cut(on), cut(tn), cut(wn), cut(mn))
        return time.perf_counter() - t0

    prev = orc.c_set_threads(cores)
    c_once(None)  # warm-up (thread pool)
    dco = min(c_once(None) for _ in range(2))
    out["c_omp"] = {"value": round(px / dco / 1e6, 4), "unit": "Mpx/s", "cores": cores, "kind": "port",
                    "sample": "best of 2 fwd + bwd of %s, oracle/pea_oracle.c with OpenMP (%d threads), %.2f s" % (what, cores, dco)}
    if shifts3d is None:
        sl = slice(0, 1)
        d1 = orc.desc_2d(en[sl], offsets)
        orc.c_set_threads(1)
        dc = c_once(sl)
        out["c_1thread"] = {"value": round(en[0, 0].size / dc / 1e6, 4), "unit": "Mpx/s", "cores": 1, "kind": "port",
                            "sample": "one fwd + bwd of 1 image, oracle/pea_oracle.c, 1 thread, %.2f s" % dc}
        torch.set_num_threads(1)
        et1, tt1, wt1, mt1 = et[:1], tt[:1], wt[:1], (None if mt is None else mt[:1])
        t0 = time.perf_counter()
        x = et1.clone().requires_grad_(True)
        orc.torch_embedding_loss(x, tt1, wt1, mt1, offsets, ema=None if emt is None else emt[:1])[0].backward()
        d1t = time.perf_counter() - t0
        torch.set_num_threads(cores)
        out["torch_1thread"] = {"value": round(en[0, 0].size / d1t / 1e6, 4), "unit": "Mpx/s", "cores": 1, "kind": "port",
                                "sample": "one fwd + bwd of 1 image, torch restatement with 1 thread, %.2f s" % d1t}
    orc.c_set_threads(prev)
    out["best_cpu_value"] = max(out["value"], out["c_omp"]["value"])
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=B_PER_GPU, help="images per GPU")
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS), help="c2 = the headline workload (default); c3 / c4 / c4n26 / c5: see CONFIGS")
    ap.add_argument("--no-train", action="store_true", help="skip the train imgs/s leg (backbone + DDP)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the default run's `configs` leg (every other BASELINE config, compact)")
    ap.add_argument("--no-section", action="store_true",
                    help="skip the multi-scale loss-section timings (profiling runs: per-kernel averages then cover the full-size launches only)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    ndev = torch.cuda.device_count()
    # one rank per GPU over RCCL; with more ranks than GPUs (a rehearsal on a smaller box) the ranks share devices and the
    # two control-plane collectives (barrier, max of the wall time) go over gloo -- there is no data-path collective
    shared = world > ndev
    torch.cuda.set_device(local % ndev)
    dev = torch.device("cuda", local % ndev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if shared:
            dist.init_process_group("gloo")
        else:
            # RCCL over xGMI.  This branch has never run on this pool (one GPU per box): a rendezvous or communicator failure must end
            # the process with a one-line reason and a non-zero code -- not hang until the driver's limit -- and is not retried in place
            import datetime
            try:
                dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=180))
                probe = torch.ones(1, device=dev)
                dist.all_reduce(probe)  # first collective: builds the communicator (rings over xGMI)
                torch.cuda.synchronize()
                if int(probe.item()) != world:
                    raise RuntimeError("all_reduce of ones over %d ranks returned %r" % (world, probe.item()))
            except Exception as ex:  # noqa: BLE001
                sys.stderr.write("bench.py: rank %d: RCCL init / first all-reduce failed: %r\n" % (rank, ex))
                sys.stderr.flush()
                os._exit(3)

    pkg = ge.load_package()
    synth = __import__("importlib").import_module(ge.PKG_NAME + ".utils.synth")
    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    if args.config != "c2":
        out = other_config(args, pkg, dev, world, rank, dist, fence)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        if out is not None:
            print(json.dumps(out), flush=True)
        return

    offsets = pkg.multi_offset(SHIFTS, NEIGHBOR)
    K, B = len(offsets), args.batch
    # each rank owns its own images (different seeds): independent units, no exchange
    e, t, w, m = synth.synth_inputs_2d(B, D, H, W, offsets, seed=555 + rank)
    E = torch.from_numpy(e).to(dev).requires_grad_(True)
    T, Wt, M = torch.from_numpy(t).to(dev), torch.from_numpy(w).to(dev), torch.from_numpy(m).to(dev)
    crit = pkg.WeightedMSE()

    def step():
        E.grad = None
        loss, affs, _ = pkg.embedding_loss(E, T, Wt, M, crit, offsets)
        pkg.backward(loss)  # loss.backward() seeded with a cached ones-scalar (no per-step fill kernel)
        return loss

    # Settle first (untimed, before the W warm-up steps): a GPU that has just been handed over idles at its lowest clocks
    # and the first few hundred launches also pay the allocator's and the library's first-touch costs; a 20-step timing
    # started cold reads 25-30 % slow.  Run batches of 20 steps until two consecutive batches agree within 2 % (<= 1.5 s).
    import gc
    gc.collect()
    settle = settle_gpu(step)
    for _ in range(args.warmup):
        step()
    # N_BATCHES timed batches of exactly --steps steps, each between barrier + synchronize, max over ranks per batch; the MEDIAN batch is
    # the headline, the fastest and the slowest are printed beside it
    bst = batch_stats(timed_batches(step, fence, args.steps, dist, "cpu" if shared else dev), args.steps)
    dt = bst["median"] * 1e-3 * args.steps
    px_per_step = world * B * H * W
    value = px_per_step * args.steps / dt / 1e6
    # the same step started with plain loss.backward(): what a drop-in caller executes -- autograd seeds a
    # scalar's backward with a ones_like fill kernel (about 5 us); reported beside the headline, never in place of it
    def step_seed():
        E.grad = None
        loss, affs, _ = pkg.embedding_loss(E, T, Wt, M, crit, offsets)
        loss.backward()

    dt_seed = wall_time_s(step_seed, fence, args.steps)
    gpu_state = gpu_state_under_load(step, dev) if rank == 0 else None  # untimed, after both timed loops
    if dist is not None:
        tmax = torch.tensor([dt_seed], dtype=torch.float64, device="cpu" if shared else dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt_seed = float(tmax.item())
    train = None
    if not args.no_train:
        try:
            train = train_leg(pkg, dev, world, rank, dist, shared, fence)
        except Exception as ex:  # the headline line must not be lost to the second metric's leg
            train = {"train_imgs_per_s": None, "train_error": repr(ex)[:300]}

    out = None
    if rank == 0:
        # ---- per-kernel durations (HIP events on the launch stream) through the C ABI --------------------
        op, L = pkg.affinity_op, pkg._lib.lib()
        spec = op.AffinitySpec(2, offsets, None, pkg._lib.BORDER_CIRCULAR, pkg._lib.NORM_BX)
        Ed = E.detach()
        desc = op.make_desc(spec, Ed)
        affs = torch.empty(B, K, H, W, device=dev)
        lossv = torch.empty(1 + K, device=dev)
        G = torch.empty(B, K, H, W, device=dev)
        wsb = L.pea_workspace_bytes(ctypes.byref(desc))
        work = torch.empty(max(wsb, 4) // 4, device=dev)
        assert L.pea_workspace_init(ctypes.c_void_p(work.data_ptr()), wsb, None) == 0  # the loss-state block: prepared once
        dE = torch.empty_like(Ed)
        one = torch.ones((), device=dev)
        P = lambda x: ctypes.c_void_p(x.data_ptr())
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        cur = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)  # (read per call: the graph capture runs on its own stream)
        INV = torch.empty(B, H, W, device=dev)  # 1 / norm plane: written by the forward, staged by the cross backward
        fwd = lambda: L.pea_affinity_fwd_ex(ctypes.byref(desc), P(Ed), None, P(T), P(Wt), P(M), P(affs), P(G), P(INV), P(lossv), P(work), wsb, cur())
        bwd = lambda: L.pea_affinity_bwd_ex2(ctypes.byref(desc), P(Ed), None, P(G), P(INV), P(affs), P(one), P(dE), None, cur())
        affs2 = torch.empty_like(affs)  # (the other entry points get their own map: `affs` is an input of the backward)
        inf = lambda: L.pea_affinity_infer(ctypes.byref(desc), P(Ed), None, P(affs2), st)
        # the labels-in training step (embedding_loss_from_labels): same outputs from the int32 label image, no t / w / m
        lab = torch.from_numpy(synth.synth_labels(B, (1, H, W), 555 + rank)[:, 0].copy()).to(dev)
        wtab = torch.empty(B * K * 2, device=dev)
        cnt_bytes = L.pea_targets_workspace_bytes(ctypes.byref(desc))
        cnt = torch.empty(cnt_bytes // 4, dtype=torch.int32, device=dev)
        lflags = pkg._lib.TGT_PADDING | pkg._lib.TGT_MASK_INSIDE
        lsb = L.pea_labels_scratch_bytes(ctypes.byref(desc))  # g + 1 / norm plane lent to the two-launch labels step
        lscr = torch.empty(max(lsb, 4) // 4, device=dev)
        labels_step = lambda: (L.pea_label_weights(ctypes.byref(desc), P(lab), lflags, P(wtab), P(cnt), cnt_bytes, st),
                               L.pea_affinity_fwd_bwd_labels_ex(ctypes.byref(desc), P(Ed), None, P(lab), P(wtab), lflags, P(affs2),
                                                                P(lossv), None, P(dE), P(work), wsb, P(lscr), lsb, st))
        labels_one = lambda: L.pea_affinity_fwd_bwd_labels(ctypes.byref(desc), P(Ed), None, P(lab), P(wtab), lflags, P(affs2),
                                                           P(lossv), None, P(dE), P(work), wsb, st)
        labels_two = lambda: L.pea_affinity_fwd_bwd_labels_ex(ctypes.byref(desc), P(Ed), None, P(lab), P(wtab), lflags, P(affs2),
                                                              P(lossv), None, P(dE), P(work), wsb, P(lscr), lsb, st)
        # the embedding head in front of the path (OutConv 32 -> D, scripts_cvppp/model/unet2d_residual.py:307): forward and
        # backward (dx, dW, db) of the 1x1 convolution on the decoder's 32-channel feature map
        HC = 32
        hx = torch.randn(B, HC, H, W, device=dev)
        hw, hb = torch.randn(D, HC, device=dev) * 0.2, torch.randn(D, device=dev)
        hdx, hdw, hdb = torch.empty_like(hx), torch.empty(D, HC, device=dev), torch.empty(D, device=dev)
        hws = L.pea_head_workspace_bytes(HC, D)
        hwork = torch.empty(hws // 4, device=dev)
        head_fwd = lambda: L.pea_head_fwd(B, HC, D, H * W, P(hx), P(hw), P(hb), P(dE), st)
        head_bwd = lambda: L.pea_head_bwd(B, HC, D, H * W, P(hx), P(hw), P(dE), P(hdx), P(hdw), P(hdb), P(hwork), hws, st)
        kt = {}
        for name, fn in (("fwd", fwd), ("bwd", bwd), ("infer", inf), ("labels_step", labels_step),
                         ("labels_one_launch", labels_one), ("labels_two_launch", labels_two),
                         ("head_fwd", head_fwd), ("head_bwd", head_bwd)):
            event_time_ms(fn, 10)
            kt[name] = event_time_ms(fn, max(20, min(args.steps, 200)))
        # ---- the training loop's loss section (five self losses over the scales + EMA cross loss + backward + relu,
        #      scripts_cvppp/main.py:284-312): call-by-call composition, one autograd node, and the labels-in path
        def section_us():
            nb_half = NEIGHBOR // 2
            labs = [torch.from_numpy(np.ascontiguousarray(synth.synth_labels(B, (1, H, W), 555 + rank)[:, 0][:, ::2 ** j, ::2 ** j])).to(dev)
                    for j in range(5)]
            embs = [torch.from_numpy(synth.synth_embedding((B, D, H >> j, W >> j), 600 + j)).to(dev) for j in range(5)]
            ema = torch.from_numpy(synth.synth_embedding((B, D, H, W), 700)).to(dev)
            tt, mm, ww = pkg.gen_targets(labs[0], offsets, padding=True)
            downs = []
            for j in range(1, 5):
                k = nb_half * (5 - j)
                tj, mj, wj = pkg.gen_targets(labs[j], offsets[:k], padding=True)
                downs.append(torch.cat([tj, wj, mj.float()], dim=1))

            def run(which):
                x = [e.detach().requires_grad_(True) for e in embs]
                if which == "labels":
                    loss, pred, _ = pkg.cvppp_loss_section_from_labels(x[0], x[1:], ema, labs[0], labs[1:], crit, offsets, nb_half,
                                                                       relu_pred=True)
                elif which == "one_node":
                    loss, pred, _ = pkg.cvppp_loss_section(x[0], x[1:], ema, tt, ww, mm, downs, crit, offsets, nb_half, relu_pred=True)
                else:
                    loss, pred, _ = pkg.cvppp_loss_section_composed(x[0], x[1:], ema, tt, ww, mm, downs, crit, offsets, nb_half)
                loss.backward()
                if which == "composed":
                    pkg.finish_pred_2d_(pred)  # F.relu(pred), main.py:312 (the other two return it already clamped)

            out = {}
            for which in ("composed", "one_node", "labels"):
                for _ in range(3):
                    run(which)
                # best of three batches: a one-off allocator stall inside a 10-call batch is not the section's cost
                out[which] = round(min(event_time_ms(lambda: run(which), 10) for _ in range(3)) * 1e3, 1)
            # the same two sections captured through the public API (pea.graphed: both streams of the section become one graph launch,
            # the ~25 launches' Python / ctypes / autograd time is gone); wall clock around 20 replays back to back
            # (composed_graphed: INTEGRATION.md section 2's drop-in sequence -- import lines only -- under pea.graphed)
            for which in ("one_node", "labels", "composed"):
                xs = [e.detach().clone().requires_grad_(True) for e in embs]

                def fn(*xs):
                    for t in xs:
                        t.grad = None
                    if which == "labels":
                        loss, pred, _ = pkg.cvppp_loss_section_from_labels(xs[0], list(xs[1:]), ema, labs[0], labs[1:], crit, offsets,
                                                                           nb_half, relu_pred=True)
                    elif which == "one_node":
                        loss, pred, _ = pkg.cvppp_loss_section(xs[0], list(xs[1:]), ema, tt, ww, mm, downs, crit, offsets, nb_half,
                                                               relu_pred=True)
                    else:
                        loss, pred, _ = pkg.cvppp_loss_section_composed(xs[0], list(xs[1:]), ema, tt, ww, mm, downs, crit, offsets, nb_half)
                    loss.backward()
                    if which == "composed":
                        pred = pkg.finish_pred_2d_(pred)
                    return (loss, pred) + tuple(t.grad for t in xs)
                out[which + "_graphed"] = graphed_us(pkg, fn, xs)
            return out

        kt_iso = {name: isolated_time_ms(fn, 20) for name, fn in (("fwd", fwd), ("bwd", bwd))}
        # the roofline uses the in-step durations (kt["fwd"] includes the loss reduction launch, as the step does)
        in_step_times_ms(fwd, bwd, 10)
        kt["fwd"], kt["bwd"], kspread = in_step_batches_ms(fwd, bwd, max(20, min(args.steps, 200)))
        section = None
        if not args.no_section:
            try:
                section = section_us()
            except Exception as ex:  # noqa: BLE001 -- an extra field: the headline line must not be lost to it
                section = {"error": repr(ex)[:300]}
        section3d = None
        if not args.no_section:
            try:
                section3d = ac3ac4_section_us(pkg, dev, rank)
            except Exception as ex:  # noqa: BLE001 -- an extra field, never the headline
                section3d = {"error": repr(ex)[:300]}
        # every other BASELINE config inside the driver's line (GPU legs only; rank 0 of a one-GPU run: the driver's BENCH command)
        configs_line = None
        if world == 1 and not args.no_configs:
            configs_line = {}
            for cn in LINE_CONFIGS:
                try:
                    configs_line[cn] = other_config(args, pkg, dev, world, rank, None, fence, cname=cn, compact=True)
                except Exception as ex:  # noqa: BLE001
                    configs_line[cn] = {"error": repr(ex)[:300]}
                torch.cuda.empty_cache()
        ab = algorithmic_bytes_per_px(D, K)
        dom = "bwd" if kt["bwd"] >= kt["fwd"] else "fwd"
        launch_bytes = ab[dom] * B * H * W
        achieved = launch_bytes / (kt[dom] * 1e-3) / 1e9
        step_gbs = ab["step"] * B * H * W / ((kt["fwd"] + kt["bwd"]) * 1e-3) / 1e9
        traffic = pmc_traffic("c2" if B == B_PER_GPU else "c2b%d" % B, dom)  # (a key per batch size: profiles/r4_final_pmc.sh)
        out = {
            "metric": "affinity-map Mpixels/sec (fwd+bwd)", "value": round(value, 2), "unit": "Mpx/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 5),
            "ms_min": round(bst["min"], 5), "ms_max": round(bst["max"], 5), "batches_ms": bst["all"],
            "timing": "median of %d batches of %d steps (each between barrier + synchronize; max over ranks per batch)" % (n_batches(args.steps), args.steps),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: CVPPP A1 embedding_loss fwd+bwd, B=%d per GPU x D=%d x %dx%d (530x500 padded), "
                                   "K=%d offsets (shifts 1,3,5,9,27 x neighbor 4), circular border, u8 mask" % (B, D, H, W, K),
                       "images_per_gpu": B, "embedding_dim": D, "height": H, "width": W, "offsets": K,
                       "sharding": "batch across ranks, no data-path collective"},
            # pixels are counted on the padded tensor the op processes (544^2 per CVPPP image); the same rate in
            # images and in pixels of the un-padded 530x500 image
            "settle_steps": settle,
            "gpu_state": gpu_state,
            "ms_per_step_autograd_seed": round(dt_seed / args.steps * 1e3, 5),
            "value_autograd_seed": round(px_per_step * args.steps / dt_seed / 1e6, 2),
            **(train or {}),
            "images_per_s_op_only": round(value * 1e6 / (H * W), 1),
            "value_530x500_equiv": round(value * (530 * 500) / (H * W), 2),
            "kernel_ms": {k: round(v, 5) for k, v in kt.items()},
            "kernel_ms_spread": kspread,  # fwd / bwd above are the medians of the same number of in-step batches
            # forward + loss finish + backward captured in a HIP graph and replayed back to back.  One step per graph pays the graph
            # launch's own fixed cost every step (MI355X_MICROARCH.md, graph-replay-floor: 10-16 us per replay, not hidden behind the
            # previous replay -- why round 4's one-step figure was SLOWER than the eager loop, whose launches queue ahead of the GPU);
            # eight steps per graph amortise it: that figure is the launch floor
            "graph_replay_ms": graph_replay_ms(fwd, bwd, max(20, min(args.steps, 200))),
            "graph_replay_x8_ms": graph_replay_ms(fwd, bwd, max(5, min(args.steps, 200) // 8), per_graph=8),
            "kernel_sum_mpx_s": round(B * H * W / ((kt["fwd"] + kt["bwd"]) * 1e-3) / 1e6, 1),
            "infer_mpx_s": round(B * H * W / (kt["infer"] * 1e-3) / 1e6, 1),
            "labels_step_mpx_s": round(B * H * W / (kt["labels_step"] * 1e-3) / 1e6, 1),
            "loss_section_us": section,
            "ac3ac4_section_us": section3d,
            "configs": configs_line,
            "roofline": {"bound": "hbm", "kernel": "pea_affinity_" + dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "algorithmic_bytes_per_px": ab[dom], "px_per_launch": B * H * W,
                         "ms": round(kt[dom], 5), "ms_drained": round(kt_iso[dom], 5),
                         "fwd_plus_bwd_GBs": round(step_gbs, 1), "fwd_plus_bwd_frac": round(step_gbs / HBM_PEAK_GBS, 4)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(offsets, e, t, w, m)
            out["speedup_vs_cpu"] = round(value / out["cpu_baseline"]["best_cpu_value"], 1)  # vs the faster CPU line
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
