"""The loss section of the reference's training loops, call for call.

The reference's drivers are out of scope, but the ORDER and SLICING of the hot-path calls inside them is what a
drop-in has to reproduce; these two functions are that section with the reference's variable names, so a maintainer
can replace scripts_cvppp/main.py:282-312 / scripts_ac3ac4/main.py:216-238 by one call.  No host synchronisation
happens inside (the reference's K `.item()` calls per loss are gone), so the section can be captured in a HIP graph
(tests/test_gpu_parity.py::test_loss_section_graph_replay).
"""
import collections.abc
import copy
import ctypes
import os

import torch

from .. import _lib
from .. import affinity_op as op
from ..loss.loss_embedding_mse import ema_embedding_loss, embedding_loss
from ..loss.loss_embedding_mse_3d import (ema_embedding_loss_norm1, ema_embedding_loss_norm5, embedding_loss_norm1,
                                          embedding_loss_norm5)
from ..utils.postproc import fill_border_relu_, relu_


def deep_weight_factor(deep_weight):
    """scripts_cvppp/main.py:214-219"""
    if deep_weight == 1:
        return [1.0, 1.0, 1.0, 1.0, 1.0]
    if deep_weight == 2:
        return [0.01, 0.03, 0.1, 0.3, 1.0]
    return [deep_weight, 1.0, 1.0, 1.0, 1.0]


_SIDE_STREAMS = {}


class _side_stream(object):
    """fork = _side_stream(dev) forks a second HIP stream of the device from the current stream AT THIS POINT; `with fork:` runs
    its launches there; `fork.join()`, called after the full-resolution kernels are enqueued on the main stream, makes the main
    stream wait for them.  The deep-supervision scales are small grids (578 tiles at 272^2 down to 8 at 34^2): one after the other
    on the full-resolution kernels' stream they cost their launch latencies, 150-200 us per section; on their own stream they fill
    the CUs the big kernels leave idle.  Capturable in a HIP graph (fork / join are stream waits).  PEA_SECTION_STREAMS=0 keeps
    everything on one stream.
    Round 6: the fork point is the constructor, not `with`: the sections enqueue the full-resolution FORWARD first and the small
    scales after it (the eager host needs 100-150 us to launch them -- profiles/r6_section3d_timeline_before.txt: the main stream's
    first kernel started 156 us after the section did -- and the GPU now works on the long kernel meanwhile); forked at `with`, the
    side stream would wait for that forward."""

    def __init__(self, dev):
        self.on = os.environ.get("PEA_SECTION_STREAMS", "1") != "0"
        self.dev = dev
        if self.on:
            self.main = torch.cuda.current_stream(self.dev)
            side = _SIDE_STREAMS.get(self.dev.index)
            if side is None:
                side = _SIDE_STREAMS[self.dev.index] = torch.cuda.Stream(device=self.dev)
            self.side = side
            side.wait_stream(self.main)  # fork: everything the callers produced is ordered before the side work

    def __enter__(self):
        if not self.on:
            return self
        self.ctx = torch.cuda.stream(self.side)
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.on:
            self.ctx.__exit__(*exc)
        return False

    def join(self):
        if self.on:
            self.main.wait_stream(self.side)  # the gradients / loss rows of the small scales are ready for what follows


class _TensorSection(torch.autograd.Function):
    """cvppp_loss_section's six losses as ONE autograd node on the tensor path: per loss one forward launch (saving g)
    and one backward launch whose dloss is the loss' weight; the EMA cross gradient is added to the self gradient with one
    add.  Same reasons as _LabelsSection: nothing per loss is left to autograd."""

    @staticmethod
    def forward(ctx, specs, weights, ema_embedding, tensors, *embs):
        ctx.set_materialize_grads(False)
        dev = embs[0].device
        L = _lib.lib()
        ncall = len(specs)
        kmax = max(sp.K for sp in specs)
        with op._on_device(dev):
            wdev = _weights_on(dev, weights)
            rows = torch.empty((ncall, 1 + kmax), dtype=torch.float32, device=dev)
            grads, pred = [], None

            def forward_one(j, e_c, o_c, want_affs, planes=0):
                """forward launch of loss j -> (desc, g, affs, inv); planes: 1 / norm planes wanted for the cross backward
                (1: self loss, 2: cross loss with the detached second operand), allocated where those kernels cover the shape"""
                spec = specs[j]
                if not want_affs and spec.act:  # a map nobody sees (the cross loss'): without the activation it can serve the backward
                    spec = copy.copy(spec)
                    spec.act = 0
                kshape = op._affs_shape(e_c, spec.K)
                t, w, m = tensors[0] if j == ncall - 1 else tensors[j]
                t, ts = op._batch_strided(t, "target", torch.float32, kshape)
                w, ws = op._batch_strided(w, "weightmap", torch.float32, kshape)
                ms = 0
                if m is not None:
                    if m.dtype == torch.bool:
                        m = m.view(torch.uint8)
                    elif m.dtype != torch.uint8:  # the reference packs the deep-supervision masks as float thirds of `downN`
                        m = m.to(torch.uint8)      # (runs on the stream of this loss: the side stream for the small scales)
                    m, ms = op._batch_strided(m, "mask", torch.uint8, kshape)
                d = op.make_desc(spec, e_c, ts, ws, ms)
                work, wsb = op.workspace(dev, d)
                g = torch.empty(kshape, dtype=torch.float32, device=dev)
                inv = None
                if planes and op.cross_supported(d, planes):
                    inv = torch.empty(((planes,) if planes == 2 else ()) + (e_c.shape[0],) + tuple(e_c.shape[2:]), dtype=torch.float32,
                                      device=dev)
                # the RAW cosine map is an input of the projection-first backward (D > 16, f16) and of the z-march backward (3D):
                # written also for a loss whose map nobody asked for, where one of those kernels takes the shape (mode 3) -- round-4
                # advice: without it the section's 3D backwards ran on the tile-per-plane kernels, not on the march
                raw_ok = inv is not None and spec.act == 0 and op.cross_supported(d, 3 if planes == 1 else 4)
                affs = torch.empty(kshape, dtype=torch.float32, device=dev) if (want_affs or raw_ok) else None
                _lib.check(L.pea_affinity_fwd_ex(ctypes.byref(d), op._ptr(e_c), op._ptr(o_c), op._ptr(t), op._ptr(w), op._ptr(m),
                                                 op._ptr(affs), op._ptr(g), op._ptr(inv), op._ptr(rows[j]), op._ptr(work), wsb,
                                                 op._stream()), "pea_affinity_fwd_ex")
                return d, g, affs, inv, (affs if raw_ok else None)

            def forward_pair(e_c, o_c):
                """the full-resolution self loss (0) and cross loss (ncall - 1) as one forward launch where the library fuses the pair
                (2D, D = 16, f32, axis-aligned stencil: csrc/pea_xdma_dual.h) -> (d0, dx, g0, gx, pred, the two 1 / norm planes) or None"""
                jx = ncall - 1
                spec0, specx = specs[0], specs[jx]
                if specx.act:  # the cross loss' map is not written
                    specx = copy.copy(specx)
                    specx.act = 0
                if spec0.K != specx.K or o_c.data_ptr() == e_c.data_ptr():
                    return None
                kshape = op._affs_shape(e_c, spec0.K)
                t, w, m = tensors[0]
                t, ts = op._batch_strided(t, "target", torch.float32, kshape)
                w, ws = op._batch_strided(w, "weightmap", torch.float32, kshape)
                ms = 0
                if m is not None:
                    if m.dtype == torch.bool:
                        m = m.view(torch.uint8)
                    elif m.dtype != torch.uint8:
                        m = m.to(torch.uint8)
                    m, ms = op._batch_strided(m, "mask", torch.uint8, kshape)
                d0 = op.make_desc(spec0, e_c, ts, ws, ms)
                if not op.cross_supported(d0, 5):
                    return None
                dx = op.make_desc(specx, e_c, ts, ws, ms)
                work, wsb = op.workspace(dev, d0, 2)
                half = wsb // 2
                g0 = torch.empty(kshape, dtype=torch.float32, device=dev)
                gx = torch.empty(kshape, dtype=torch.float32, device=dev)
                affs = torch.empty(kshape, dtype=torch.float32, device=dev)
                inv = torch.empty((2, e_c.shape[0]) + tuple(e_c.shape[2:]), dtype=torch.float32, device=dev)
                rc = L.pea_affinity_fwd_dual_ex(ctypes.byref(d0), ctypes.byref(dx), op._ptr(e_c), op._ptr(o_c), op._ptr(t), op._ptr(w), op._ptr(m),
                                                op._ptr(affs), op._ptr(g0), op._ptr(gx), op._ptr(inv[0]), op._ptr(inv[1]), op._ptr(rows[0]),
                                                op._ptr(rows[jx]), ctypes.c_void_p(work.data_ptr()), ctypes.c_void_p(work.data_ptr() + half), half,
                                                op._stream())
                if rc == _lib.E_UNSUPPORTED:  # (nothing was launched)
                    return None
                _lib.check(rc, "pea_affinity_fwd_dual_ex")
                return d0, dx, g0, gx, affs, inv

            def backward_one(j, d, e_c, o_c, g, inv=None, de=None, raw=None):
                """raw: the forward's map, untouched (the backward runs inside this node's forward, before `pred` is handed out)"""
                de = torch.empty_like(e_c) if de is None else de
                rc = L.pea_affinity_bwd_ex2(ctypes.byref(d), op._ptr(e_c), op._ptr(o_c), op._ptr(g), op._ptr(inv), op._ptr(raw),
                                            op._ptr(wdev[j:j + 1]), op._ptr(de), None, op._stream())
                if rc == _lib.E_UNSUPPORTED:
                    return None
                _lib.check(rc, "pea_affinity_bwd_ex2")
                return de

            # ---- full resolution: self + cross; their backwards are ONE launch: the cross kernel with a second phase where the
            #      shape allows (pea_affinity_bwd_dual_ex with the 1 / norm planes the two forwards wrote: 2D, D = 16, axis-aligned
            #      stencil), else the tiled two-phase kernel, else two launches and an add
            jx = ncall - 1
            small = []
            fork = _side_stream(dev)  # (the fork point: the small scales wait for nothing this section launches)
            e0 = op._embedding_arg(embs[0], "embedding")
            ema_c = op._embedding_arg(ema_embedding, "ema_embedding").to(e0.dtype)
            pair = forward_pair(e0, ema_c)
            if pair is not None:  # one forward launch for the two losses (pea_affinity_fwd_dual_ex: e, target, weight, mask read once)
                d0, dxx, g0, gx, pred, invx = pair  # invx: the two planes (own, second operand's) as the cross forward writes them
                inv0, inv_other, raw0, rawx = invx[0], invx[1], None, None
            else:
                d0, g0, pred, inv0, raw0 = forward_one(0, e0, None, True, 1)
                dxx, gx, _, invx, rawx = forward_one(jx, e0, ema_c, False, 2 if inv0 is not None else 0)
                inv_other = None if invx is None else invx[1]
            with fork:  # the deep-supervision scales, on their own stream beside the full-resolution pair
                for j in range(1, jx):
                    e_c = op._embedding_arg(embs[j], "embedding")
                    # (the 1 / norm plane goes along wherever the cross backward takes the scale -- 272^2 down to 68^2 -- : round 4 kept
                    #  the small scales on the tiled backward "because their grids are launch-sized"; measured in round 5 the cross
                    #  backward is worth 36 us of the section, profiles/r5_section_small.txt.  No raw map: D = 16 reads none.)
                    d, g, _, inv, raw = forward_one(j, e_c, None, False, 1)
                    small.append(backward_one(j, d, e_c, None, g, inv, raw=raw))
            de0 = torch.empty_like(e0)
            rc = L.pea_affinity_bwd_dual_ex(ctypes.byref(d0), op._ptr(e0), op._ptr(ema_c), op._ptr(g0), op._ptr(gx), op._ptr(inv0),
                                            op._ptr(inv_other), op._ptr(wdev[0:1]), op._ptr(wdev[jx:jx + 1]),
                                            op._ptr(de0), op._stream())
            if rc == _lib.E_UNSUPPORTED:
                de0 = backward_one(0, d0, e0, None, g0, inv0, raw=raw0)
                # the cross loss' gradient ADDED by its kernel (PEA_FLAG_ACCUMULATE_DE: the role-A cross backward, 2D and 3D) -- else a
                # second buffer and an add over [B,D,...] (27 us of the 3D section, profiles/r6_section3d_timeline_before.txt)
                if backward_one(jx, _accumulating(dxx), e0, ema_c, gx, invx, de=de0, raw=rawx) is None:
                    de0.add_(backward_one(jx, dxx, e0, ema_c, gx, invx, raw=rawx))
            else:
                _lib.check(rc, "pea_affinity_bwd_dual_ex")
            grads.append(de0)
            grads.extend(small)
            fork.join()
            losses = rows[:, 0]
            total = _section_total(L, rows, wdev, ncall)
        ctx.grads, ctx.n_embs = grads, len(embs)
        _stash_again(ctx, _TensorSection, (specs, weights, ema_embedding, tensors) + tuple(embs))
        ctx.mark_non_differentiable(pred, losses)
        return total, pred, losses

    backward = staticmethod(lambda ctx, dtotal, _dp, _dl: _section_backward(ctx, dtotal))


_ACC_DESC = {}


def _accumulating(d):
    """the memoised descriptor d with PEA_FLAG_ACCUMULATE_DE set (pea_affinity_bwd_ex2 then adds to de)"""
    hit = _ACC_DESC.get(id(d))
    if hit is None or hit[0] is not d:
        if len(_ACC_DESC) > 256:
            _ACC_DESC.clear()
        da = type(d).from_buffer_copy(d)
        da.flags |= _lib.FLAG_ACCUMULATE_DE
        hit = _ACC_DESC[id(d)] = (d, da)  # (d is kept: its id cannot be reused while the entry lives)
    return hit[1]


def _section_total(L, rows, wdev, ncall):
    """sum_j weight_j * loss_j (scripts_cvppp/main.py:295-306) from the loss rows of the section's calls: one small launch
    (pea_weighted_sum) instead of a multiply and a reduction -- they sit at the end of the section's critical path, 5 us each"""
    total = torch.empty((), dtype=torch.float32, device=rows.device)
    _lib.check(L.pea_weighted_sum(op._ptr(rows), rows.stride(0), op._ptr(wdev), ncall, op._ptr(total), op._stream()), "pea_weighted_sum")
    return total


def _stash_again(ctx, cls, args):
    """keep a section's arguments for a second backward over a retained graph: every tensor goes through save_for_backward (freed
    after a non-retained backward, version-checked on a retained one -- a plain ctx attribute would keep ~150 MB of embeddings alive
    until the loss tensor is dropped and miss in-place edits); the nesting and the non-tensor values stay on ctx"""
    flat = []

    def enc(x):
        if isinstance(x, torch.Tensor):
            flat.append(x)
            return ("T", len(flat) - 1)
        if isinstance(x, (list, tuple)):
            return ("L" if isinstance(x, list) else "U", [enc(v) for v in x])
        if isinstance(x, dict):
            return ("D", [(k, enc(v)) for k, v in x.items()])
        if isinstance(x, _LabelCfg):
            return ("C", enc([x.labels, x.lflags, x.gen_flags, x.has_mask, x.tables]))
        return ("V", x)

    ctx.again = (cls, enc(tuple(args)))
    ctx.save_for_backward(*flat)


def _unstash_again(ctx):
    flat = ctx.saved_tensors

    def dec(t):
        kind, v = t
        if kind == "T":
            return flat[v]
        if kind == "L":
            return [dec(i) for i in v]
        if kind == "U":
            return tuple(dec(i) for i in v)
        if kind == "D":
            return {k: dec(i) for k, i in v}
        if kind == "C":
            return _LabelCfg(*dec(v))
        return v

    cls, tree = ctx.again
    return cls, dec(tree)


class _Rerun(object):
    """stands in for the autograd context when a section is computed again for a second backward over a retained graph"""

    def set_materialize_grads(self, value):
        pass

    def save_for_backward(self, *tensors):
        pass

    def mark_non_differentiable(self, *tensors):
        pass


def _section_backward(ctx, dtotal):
    n = ctx.n_embs
    if dtotal is None:
        return (None, None, None, None) + (None,) * n
    if ctx.grads is None:
        # the first backward took the gradient buffers (scaled in place): a second backward over a retained graph computes the
        # section again on the saved inputs -- the rare case pays, not every step
        cls, args = _unstash_again(ctx)
        stub = _Rerun()
        cls.forward(stub, *args)
        ctx.grads = stub.grads
    grads, ctx.grads = ctx.grads, None
    L = _lib.lib()
    dev = grads[0].device
    with op._on_device(dev):
        # one launch rescales every gradient by grad_output (and returns untouched when that is exactly 1); pea.backward(loss) seeds
        # the backward with ITS cached ones-scalar: recognised by identity, the launch is skipped
        if any(dtotal is one for one in tuple(op._ONES.values())):  # (a snapshot: another thread's pea.backward may add a device's entry)
            return (None, None, None, None) + tuple(g if ctx.needs_input_grad[4 + k] else None for k, g in enumerate(grads))
        dl = dtotal.to(device=dev, dtype=torch.float32).contiguous()
        bufs = (ctypes.c_void_p * n)(*[g.data_ptr() for g in grads])
        cnts = (ctypes.c_size_t * n)(*[g.numel() for g in grads])
        _lib.check(L.pea_scale_inplace_multi(bufs, cnts, n, _lib.F16 if grads[0].dtype == torch.float16 else _lib.F32, op._ptr(dl),
                                             op._stream()), "pea_scale_inplace_multi")
    return (None, None, None, None) + tuple(g if ctx.needs_input_grad[4 + k] else None for k, g in enumerate(grads))


_SPEC_CACHE = {}


def _section_specs(offsets, nb_half, affs0_weight, dis_mode, deep_weight, self_emb, cross_emb):
    """the six AffinitySpecs and loss weights of the section; built once per configuration (the callers get shallow
    copies: they set relu / label flags on theirs)"""
    key = (tuple(tuple(o) for o in offsets), nb_half, affs0_weight, dis_mode, deep_weight, self_emb, cross_emb)
    hit = _SPEC_CACHE.get(key)
    if hit is None:
        if len(_SPEC_CACHE) > 64:
            _SPEC_CACHE.clear()
        hit = _SPEC_CACHE[key] = _build_section_specs(offsets, nb_half, affs0_weight, dis_mode, deep_weight, self_emb, cross_emb)
    return [copy.copy(sp) for sp in hit[0]], hit[1]


def _build_section_specs(offsets, nb_half, affs0_weight, dis_mode, deep_weight, self_emb, cross_emb):
    from ..loss.loss_embedding_mse import _spec
    dwf = deep_weight_factor(deep_weight)
    specs = [_spec(offsets, [1.0] * len(offsets), dis_mode)]
    for j in range(4):
        k = nb_half * (4 - j)
        specs.append(_spec(offsets[:k], [1.0] * k, dis_mode))
    specs.append(_spec(offsets, [float(affs0_weight) if i < 2 else 1.0 for i in range(len(offsets))], dis_mode))
    weights = [dwf[0] * self_emb] + [dwf[j + 1] * self_emb for j in range(4)] + [dwf[0] * cross_emb]
    return specs, weights


class _SectionParts(collections.abc.Mapping):
    """{"loss_embedding", "loss_emd": [4], "loss_embedding_cross"}: the individual weighted losses the reference's loop logs
    (main.py:298-309), evaluated on first access -- seven 5 us elementwise launches that a step which only calls
    loss.backward() never needs"""
    _KEYS = ("loss_embedding", "loss_emd", "loss_embedding_cross")

    def __init__(self, losses, weights, self_emb, cross_emb):
        self._args, self._d = (losses, weights, self_emb, cross_emb), None

    def _get(self):
        if self._d is None:
            losses, weights, self_emb, cross_emb = self._args
            wl = losses * _weights_on(losses.device, weights)
            self._d = {"loss_embedding": wl[0] / self_emb if self_emb else wl[0],
                       "loss_emd": [wl[1 + j] / self_emb if self_emb else wl[1 + j] for j in range(4)],
                       "loss_embedding_cross": wl[5] / cross_emb if cross_emb else wl[5]}
        return self._d

    def __getitem__(self, k):
        return self._get()[k]

    def __iter__(self):
        return iter(self._KEYS)

    def __len__(self):
        return len(self._KEYS)


def _section_parts(losses, weights, self_emb, cross_emb):
    return _SectionParts(losses, weights, self_emb, cross_emb)


def cvppp_loss_section(embedding, emds, ema_embedding, target, weightmap, affs_mask, downs, criterion, offsets, nb_half,
                       affs0_weight=1, dis_mode='ours', deep_weight=1, self_emb=1.0, cross_emb=1.0, relu_pred=False):
    """scripts_cvppp/main.py:284-310 (and scripts_bbbc/main.py:279-305): five self losses over the deep-supervision
    scales + the EMA cross loss at full resolution.

    emds = (emd1, emd2, emd3, emd4); downs = (down1, down2, down3, down4), each packed [B, 3k, h, w] =
    (target | weight | mask) thirds with k = nb_half * (4, 3, 2, 1) channels.  ema_embedding is the flipped-back,
    detached EMA output (convert_consistency_flip).  Returns (loss without the consistency term `loss_mask`, pred,
    parts) where pred is the full-resolution affinity map BEFORE relu (call relu_ after backward like the
    reference does at :312) and parts the individual weighted losses (device scalars).

    With the fused criterion (WeightedMSE) and a detached EMA operand the six losses run as one autograd node
    (_TensorSection); otherwise they are composed from embedding_loss / ema_embedding_loss call by call.

    relu_pred=True: pred comes back already clamped at 0 -- the kernel that writes the map applies the reference's next
    statement, `pred = F.relu(pred)` (:312), on the way out, so that line (finish_pred_2d_) and its pass over
    [B,K,H,W] are dropped; nothing else reads pred in the training loop."""
    if getattr(criterion, 'pea_fused', False) and not ema_embedding.requires_grad:
        specs, weights = _section_specs(offsets, nb_half, affs0_weight, dis_mode, deep_weight, self_emb, cross_emb)
        specs[0].relu = specs[-1].relu = bool(relu_pred)  # (the cross loss writes no map; its descriptor must match)
        tensors = [(target, weightmap, affs_mask)]
        for j, down in enumerate(downs):
            k = nb_half * (4 - j)
            m = down[:, 2 * k:3 * k]
            tensors.append((down[:, 0:k], down[:, k:2 * k], m))  # (a float mask third is converted where its loss is launched: side stream)
        loss, pred, losses = _TensorSection.apply(specs, weights, ema_embedding, tensors, embedding, *emds)
        return loss, pred, _section_parts(losses, weights, self_emb, cross_emb)
    loss, pred, parts = cvppp_loss_section_composed(embedding, emds, ema_embedding, target, weightmap, affs_mask, downs, criterion,
                                                    offsets, nb_half, affs0_weight, dis_mode, deep_weight, self_emb, cross_emb)
    # (out of place: the composed path's `pred` may be saved for its loss' backward -- the projection-first kernel reads the raw map)
    return loss, (torch.relu(pred) if relu_pred else pred), parts


def cvppp_loss_section_composed(embedding, emds, ema_embedding, target, weightmap, affs_mask, downs, criterion, offsets, nb_half,
                                affs0_weight=1, dis_mode='ours', deep_weight=1, self_emb=1.0, cross_emb=1.0):
    """the same section call by call, statement for statement as scripts_cvppp/main.py:284-310 (any criterion)"""
    dwf = deep_weight_factor(deep_weight)
    losses = []
    for j, (emd, down) in enumerate(zip(emds, downs)):
        k = nb_half * (4 - j)
        l, _, _ = embedding_loss(emd, down[:, 0:k], down[:, k:2 * k], down[:, 2 * k:3 * k], criterion, offsets[:k],
                                 affs0_weight=affs0_weight, mode=dis_mode)
        losses.append(l)
    loss_embedding, pred, _ = embedding_loss(embedding, target, weightmap, affs_mask, criterion, offsets,
                                             affs0_weight=affs0_weight, mode=dis_mode)
    loss_embedding_cross, _ = ema_embedding_loss(embedding, ema_embedding, target, weightmap, affs_mask, criterion, offsets,
                                                 affs0_weight=affs0_weight, mode=dis_mode)
    loss_embedding = loss_embedding * dwf[0]
    loss_emd = [losses[j] * dwf[j + 1] for j in range(4)]
    loss_embedding_cross = loss_embedding_cross * dwf[0]
    loss_embedding_total = (loss_emd[0] + loss_emd[1] + loss_emd[2] + loss_emd[3] + loss_embedding) * self_emb
    loss_embedding_cross_total = loss_embedding_cross * cross_emb
    loss = loss_embedding_total + loss_embedding_cross_total
    parts = {"loss_embedding": loss_embedding, "loss_emd": loss_emd, "loss_embedding_cross": loss_embedding_cross}
    return loss, pred, parts


def _specs_3d(embedding_mode, affs0_weight):
    from ..loss.loss_embedding_mse_3d import _spec as spec3
    from ..utils.affinity_ours import NORM5_SHIFTS
    if embedding_mode == 1:
        full = ([1, 1, 1], 1)
    elif embedding_mode == 5:
        full = (NORM5_SHIFTS, 3)
    else:
        raise NotImplementedError
    specs = [spec3(full[0], affs0_weight, full[1])] + [spec3([1, 1, 1], affs0_weight, 1) for _ in range(4)] + [spec3(full[0], affs0_weight, full[1])]
    return specs, [1.0] * 6


def ac3ac4_loss_section(embedding, emds, ema_embedding, target, weightmap, downs, criterion, embedding_mode=5, affs0_weight=1,
                        finish_pred=False):
    """scripts_ac3ac4/main.py:219-231: full-resolution self + EMA cross loss (norm1 or norm5) and four norm1 losses on
    the deep-supervision heads; downs = (down1, .., down4) packed [B, 6, z, y, x] = (target[:3] | weight[3:]),
    paired emd1<->down4 .. emd4<->down1 as in the reference.  Returns (loss, pred before the border fill / relu);
    call finish_pred_3d_(pred) after backward (:233-237).  One autograd node with the fused criterion and a detached
    EMA operand, the call-by-call composition otherwise.

    finish_pred=True: pred comes back FINISHED -- the reference's next five statements (:233-237: border fill of the three shift-1
    channels, F.relu) applied: where no backward kernel reads the raw map (the reference's training crops) the forward clamps the
    map on the way out and only the border slices are touched afterwards (pea_fill_border_relu with relu = 0: 5 us instead of a pass
    over [B,12,Z,Y,X]); else one fill + relu pass.  Do not call finish_pred_3d_ again (it would be harmless: both are idempotent)."""
    if getattr(criterion, 'pea_fused', False) and not ema_embedding.requires_grad and embedding_mode in (1, 5):
        specs, weights = _specs_3d(embedding_mode, affs0_weight)
        tensors = [(target, weightmap, None)] + [(d[:, :3], d[:, 3:], None) for d in downs[::-1]]
        clamped = False
        if finish_pred and embedding.is_cuda:
            # (the z-march backward of large volumes reads the forward's RAW map: pea_cross_supported mode 3)
            clamped = not op.cross_supported(op.make_desc(specs[0], op._embedding_arg(embedding, "embedding")), 3)
            specs[0].relu = clamped
        loss, pred, _ = _TensorSection.apply(specs, weights, ema_embedding, tensors, embedding, *emds)
        if finish_pred:
            fill_border_relu_(pred, shift=1, relu=not clamped)
        return loss, pred
    loss, pred = ac3ac4_loss_section_composed(embedding, emds, ema_embedding, target, weightmap, downs, criterion, embedding_mode, affs0_weight)
    return loss, (finish_pred_3d_(pred.detach().clone()) if finish_pred else pred)


def ac3ac4_loss_section_from_labels(embedding, emds, ema_embedding, labels, label_downs, criterion, embedding_mode=5, affs0_weight=1):
    """ac3ac4_loss_section from the segmentation: labels [B,Z,Y,X] and label_downs = (seg of down1, .., seg of down4) replace
    target / weightmap / down1..4 (seg_to_aff(pad='') + weight_binary_ratio are evaluated inside the kernels)."""
    if not getattr(criterion, 'pea_fused', False):
        raise NotImplementedError("the labels-in section fuses WeightedMSE")
    if ema_embedding.requires_grad:
        raise NotImplementedError("the EMA operand must be detached (convert_consistency_flip)")
    specs, weights = _specs_3d(embedding_mode, affs0_weight)
    label_cfg = _LabelCfg([labels] + list(label_downs[::-1]), _lib.TGT_BOTH_FOREGROUND, _lib.TGT_BOTH_FOREGROUND, False, None)
    loss, pred, _ = _LabelsSection.apply(specs, weights, ema_embedding, label_cfg, embedding, *emds)
    return loss, pred


def ac3ac4_loss_section_composed(embedding, emds, ema_embedding, target, weightmap, downs, criterion, embedding_mode=5, affs0_weight=1):
    """the same section call by call, statement for statement as scripts_ac3ac4/main.py:219-231 (any criterion)"""
    if embedding_mode == 1:
        loss_embedding, pred = embedding_loss_norm1(embedding, target, weightmap, criterion, affs0_weight=affs0_weight)
        loss_embedding_cross, _ = ema_embedding_loss_norm1(embedding, ema_embedding, target, weightmap, criterion,
                                                           affs0_weight=affs0_weight)
    elif embedding_mode == 5:
        loss_embedding, pred = embedding_loss_norm5(embedding, target, weightmap, criterion, affs0_weight=affs0_weight)
        loss_embedding_cross, _ = ema_embedding_loss_norm5(embedding, ema_embedding, target, weightmap, criterion,
                                                           affs0_weight=affs0_weight)
    else:
        raise NotImplementedError
    loss = loss_embedding + loss_embedding_cross
    for emd, down in zip(emds, downs[::-1]):
        l, _ = embedding_loss_norm1(emd, down[:, :3], down[:, 3:], criterion, affs0_weight=affs0_weight)
        loss = loss + l
    return loss, pred


def finish_pred_3d_(pred, shift=1):
    """scripts_ac3ac4/main.py:233-237: border fill of the three shift-1 channels, then relu; in place"""
    return fill_border_relu_(pred, shift=shift, relu=True)


def cvppp_validation_section(embedding, emds, target, weightmap, affs_mask, downs, criterion, offsets, nb_half,
                             affs0_weight=1, dis_mode='ours', test_mode=False):
    """The validation / test caller of scripts_cvppp/inference.py:179-193 (and the validation branch of main.py:380-395):
    under no_grad, either embedding2affs alone (mode == 'test') or the five self losses -- unweighted sum, as the reference
    adds them at :190 -- and the full-resolution map; `pred` comes back as F.relu(pred) (:193).  Returns (loss or None,
    pred); the caller adds its own loss_mask term.  emds = (emd1, .., emd4), downs = (down1, .., down4) as in training."""
    with torch.no_grad():
        if test_mode:
            from ..loss.loss_embedding_mse import embedding2affs
            return None, finish_pred_2d_(embedding2affs(embedding, offsets, mode=dis_mode))
        total = None
        fork = _side_stream(embedding.device)
        with fork:  # the small scales beside the full-resolution call, as in training
            for j, (emd, down) in enumerate(zip(emds, downs)):
                k = nb_half * (4 - j)
                l, _, _ = embedding_loss(emd, down[:, 0:k], down[:, k:2 * k], down[:, 2 * k:3 * k], criterion, offsets[:k],
                                         affs0_weight=affs0_weight, mode=dis_mode)
                total = l if total is None else total + l
        loss_embedding, pred, _ = embedding_loss(embedding, target, weightmap, affs_mask, criterion, offsets,
                                                 affs0_weight=affs0_weight, mode=dis_mode)
        fork.join()
        return total + loss_embedding, finish_pred_2d_(pred)


def finish_pred_2d_(pred):
    """scripts_cvppp/main.py:312"""
    return relu_(pred)


class _LabelCfg(object):
    """what the labels-in section needs besides the embeddings: the label images (full resolution first), the target flags
    of the fused kernels / of pea_gen_targets (the fallback for scales smaller than a tile), whether a mask exists, and
    optionally the class-balance tables computed ahead"""

    def __init__(self, labels, lflags, gen_flags, has_mask, tables):
        self.labels, self.lflags, self.gen_flags, self.has_mask, self.tables = labels, lflags, gen_flags, has_mask, tables


class _LabelsSection(torch.autograd.Function):
    """The whole loss section as ONE autograd node: every loss is a labels-in launch that writes its gradient already
    multiplied by its weight (deep_weight_factor x self_emb / cross_emb is known before the launch), the EMA cross loss
    accumulates onto the self loss' gradient in the kernel's epilogue, and the weighted total is two tiny ops on a
    [6]-vector of device scalars.  Nothing is left for autograd to do per loss: no scalar-multiply kernels, no gradient
    accumulation pass over [B,D,H,W], no zero-filled gradients for the affinity maps (40 % of the section's GPU time
    when it is composed from the per-loss functions, profiles/r1c_loss_section.txt)."""

    @staticmethod
    def forward(ctx, specs, weights, ema_embedding, label_cfg, *embs):
        ctx.set_materialize_grads(False)
        dev = embs[0].device
        L = _lib.lib()
        labels_list, lflags, gen_flags, has_mask = label_cfg.labels, label_cfg.lflags, label_cfg.gen_flags, label_cfg.has_mask
        tables = label_cfg.tables  # precomputed by *_label_weight_tables (off the critical path), or None
        ncall = len(specs)
        kmax = max(sp.K for sp in specs)
        with op._on_device(dev):
            wdev = _weights_on(dev, weights)
            rows = torch.empty((ncall, 1 + kmax), dtype=torch.float32, device=dev)

            def prep(j):
                e_c = op._embedding_arg(embs[j], "embedding")
                lab = op._labels_int32(labels_list[j])
                d = op.make_desc(specs[j], e_c)
                cb = L.pea_targets_workspace_bytes(ctypes.byref(d))
                counts = torch.empty(max(cb, 4) // 4, dtype=torch.int32, device=dev)
                if tables is not None:
                    wtab = tables[j]
                    if wtab.numel() != e_c.shape[0] * specs[j].K * 2 or wtab.dtype != torch.float32 or wtab.device != dev:
                        raise ValueError("weight table %d does not fit this batch / stencil" % j)
                    return e_c, lab, d, wtab, counts, cb
                wtab = torch.empty(e_c.shape[0] * specs[j].K * 2, dtype=torch.float32, device=dev)
                _lib.check(L.pea_label_weights(ctypes.byref(d), op._ptr(lab), lflags, op._ptr(wtab), op._ptr(counts), cb, op._stream()),
                           "pea_label_weights")
                return e_c, lab, d, wtab, counts, cb

            def one(j, e_c, o_c, lab, d, wtab, counts, cb, affs, de, accumulate):
                """loss j as one labels-in launch (or, where no labels kernel applies -- the coarsest scales are smaller
                than a tile --, targets on the GPU + the two tensor launches), gradient weighted by weights[j]"""
                work, wsb = op.workspace(dev, d)
                fl = lflags | (_lib.TGT_ACCUMULATE if accumulate else 0)
                rc = L.pea_affinity_fwd_bwd_labels(ctypes.byref(d), op._ptr(e_c), op._ptr(o_c), op._ptr(lab), op._ptr(wtab), fl,
                                                   op._ptr(affs), op._ptr(rows[j]), op._ptr(wdev[j:j + 1]), op._ptr(de), op._ptr(work),
                                                   wsb, op._stream())
                if rc != _lib.E_UNSUPPORTED:
                    _lib.check(rc, "pea_affinity_fwd_bwd_labels")
                    return
                if accumulate:
                    raise NotImplementedError("the cross loss needs the labels-in kernel (D = 16 or 32)")
                kshape = op._affs_shape(e_c, specs[j].K)
                t = torch.empty(kshape, dtype=torch.float32, device=dev)
                m = torch.empty(kshape, dtype=torch.uint8, device=dev) if has_mask else None
                w = torch.empty(kshape, dtype=torch.float32, device=dev)
                _lib.check(L.pea_gen_targets(ctypes.byref(d), op._ptr(lab), gen_flags, op._ptr(t), op._ptr(m), op._ptr(w),
                                             op._ptr(counts), cb, op._stream()), "pea_gen_targets")
                g = torch.empty(kshape, dtype=torch.float32, device=dev)
                _lib.check(L.pea_affinity_fwd(ctypes.byref(d), op._ptr(e_c), op._ptr(o_c), op._ptr(t), op._ptr(w), op._ptr(m), op._ptr(affs),
                                              op._ptr(g), op._ptr(rows[j]), op._ptr(work), wsb, op._stream()), "pea_affinity_fwd")
                _lib.check(L.pea_affinity_bwd(ctypes.byref(d), op._ptr(e_c), op._ptr(o_c), op._ptr(g), op._ptr(wdev[j:j + 1]), op._ptr(de),
                                              None, op._stream()), "pea_affinity_bwd")

            # ---- full resolution: self + cross.  One launch with two LDS phases when the library has it (same labels,
            #      same weights, same own pixel), else two launches, the second accumulating onto the first's gradient
            jx = ncall - 1
            small = []
            fork = _side_stream(dev)  # (the fork point)
            e0, lab0, d0, wtab0, counts0, cb0 = prep(0)
            ema_c = op._embedding_arg(ema_embedding, "ema_embedding").to(e0.dtype)
            dx_ = op.make_desc(specs[jx], e0)
            pred = torch.empty(op._affs_shape(e0, specs[0].K), dtype=torch.float32, device=dev)
            de0 = torch.empty_like(e0)
            work2, wsb2 = op.workspace(dev, d0, 2)
            rc = L.pea_affinity_fwd_bwd_labels_dual(ctypes.byref(d0), ctypes.byref(dx_), op._ptr(e0), op._ptr(ema_c), op._ptr(lab0),
                                                    op._ptr(wtab0), lflags, op._ptr(pred), op._ptr(rows[0]), op._ptr(rows[jx]),
                                                    op._ptr(wdev[0:1]), op._ptr(wdev[jx:jx + 1]), op._ptr(de0), op._ptr(work2), wsb2, op._stream())
            if rc == _lib.E_UNSUPPORTED:
                one(0, e0, None, lab0, d0, wtab0, counts0, cb0, pred, de0, False)
                one(jx, e0, ema_c, lab0, dx_, wtab0, counts0, cb0, None, de0, True)
            else:
                _lib.check(rc, "pea_affinity_fwd_bwd_labels_dual")
            with fork:  # the deep-supervision scales, on their own stream beside the full-resolution pair (enqueued behind it: _side_stream)
                for j in range(1, jx):
                    e_c, lab, d, wtab, counts, cb = prep(j)
                    de = torch.empty_like(e_c)
                    one(j, e_c, None, lab, d, wtab, counts, cb, None, de, False)
                    small.append(de)
            grads = [de0] + small
            fork.join()
            losses = rows[:, 0]
            total = _section_total(L, rows, wdev, ncall)
        ctx.grads, ctx.n_embs = grads, len(embs)
        _stash_again(ctx, _LabelsSection, (specs, weights, ema_embedding, label_cfg) + tuple(embs))
        ctx.mark_non_differentiable(pred, losses)
        return total, pred, losses

    backward = staticmethod(lambda ctx, dtotal, _dp, _dl: _section_backward(ctx, dtotal))


_WEIGHT_CACHE = {}


def _weights_on(dev, weights):
    """device copy of a tuple of Python floats, made once per distinct tuple (no H2D copy inside the step)"""
    key = (str(dev), tuple(float(w) for w in weights))
    t = _WEIGHT_CACHE.get(key)
    if t is None:
        t = torch.tensor(key[1], dtype=torch.float32, device=dev)
        _WEIGHT_CACHE[key] = t
    return t


def cvppp_label_weight_tables(labels, label_downs, offsets, nb_half, dis_mode='ours'):
    """The class-balance weight tables (weight_binary_ratio per image and channel, scripts_cvppp/data/data_segmentation.py:
    205-228) of the five scales, straight from the label images.  They depend on the labels only: call this as soon as the
    label batch is on the GPU -- e.g. before the backbone's forward -- and hand the result to
    cvppp_loss_section_from_labels(weight_tables=...), which then starts with the loss kernels instead of the count
    reductions (33 us at full resolution on the critical path otherwise)."""
    specs, _ = _section_specs(offsets, nb_half, 1, dis_mode, 1, 1.0, 1.0)
    flags = _lib.TGT_PADDING | _lib.TGT_MASK_INSIDE
    L = _lib.lib()
    tables = []
    for j, lab in enumerate([labels] + list(label_downs)):
        op._require_gpu(lab, "labels")
        lab = op._labels_int32(lab)
        like = torch.empty((lab.shape[0], 16) + tuple(lab.shape[1:]), device="meta")  # geometry only
        d = op.make_desc(specs[j], like)
        with op._on_device(lab.device):
            cb = L.pea_targets_workspace_bytes(ctypes.byref(d))
            counts = torch.empty(max(cb, 4) // 4, dtype=torch.int32, device=lab.device)
            wtab = torch.empty(lab.shape[0] * specs[j].K * 2, dtype=torch.float32, device=lab.device)
            _lib.check(L.pea_label_weights(ctypes.byref(d), op._ptr(lab), flags, op._ptr(wtab), op._ptr(counts), cb, op._stream()),
                       "pea_label_weights")
        tables.append(wtab)
    return tables


def cvppp_loss_section_from_labels(embedding, emds, ema_embedding, labels, label_downs, criterion, offsets, nb_half,
                                   affs0_weight=1, dis_mode='ours', deep_weight=1, self_emb=1.0, cross_emb=1.0, relu_pred=False,
                                   weight_tables=None):
    """cvppp_loss_section without any target / weight / mask tensor: `labels` [B,H,W] and `label_downs` = the four
    nearest-downsampled label images (scripts_cvppp/data/data_provider.py:199-208) replace target, weightmap, affs_mask
    and down1..down4 (gen_affs_ours(padding=True) + weight_binary_ratio are evaluated inside the kernels).  The six
    losses run as one autograd node (_LabelsSection).  Returns (loss, pred, parts) like cvppp_loss_section; the entries
    of parts are the weighted per-loss values (device scalars, no gradient of their own).  relu_pred: as there.
    weight_tables: the result of cvppp_label_weight_tables for this label batch, computed earlier in the step."""
    if not getattr(criterion, 'pea_fused', False):
        raise NotImplementedError("the labels-in section fuses WeightedMSE; use cvppp_loss_section for another criterion")
    if ema_embedding.requires_grad:
        raise NotImplementedError("the EMA operand must be detached (convert_consistency_flip)")
    specs, weights = _section_specs(offsets, nb_half, affs0_weight, dis_mode, deep_weight, self_emb, cross_emb)
    specs[0].relu = specs[-1].relu = bool(relu_pred)
    label_cfg = _LabelCfg([labels] + list(label_downs), _lib.TGT_PADDING | _lib.TGT_MASK_INSIDE, _lib.TGT_PADDING, True,
                          None if weight_tables is None else list(weight_tables))  # gen_affs_ours(padding=True) + its mask
    loss, pred, losses = _LabelsSection.apply(specs, weights, ema_embedding, label_cfg, embedding, *emds)
    return loss, pred, _section_parts(losses, weights, self_emb, cross_emb)
