"""Host side of the hot path: descriptor construction, argument checking, autograd glue.

PyTorch is plumbing here (device memory, streams, autograd bookkeeping); all arithmetic happens in
libpea_hip.so through the C ABI of include/pea.h.  Nothing in this file has a CPU implementation:
tensors that are not on a ROCm device raise, and a missing library raises PeaLibraryError.

Reference behaviour mirrored (file:line in the reference tree):
  * embedding_loss / ema_embedding_loss / embedding2affs      scripts_cvppp/loss/loss_embedding_mse.py:18-95
  * embedding_loss_norm1/5, ema_*, inf_*                        scripts_ac3ac4/loss/loss_embedding_mse.py:7-289
  * WeightedMSE normaliser                                      scripts_cvppp/loss/loss.py:112-119
"""
import collections
import collections.abc
import ctypes
import os
import threading

import torch

from . import _lib
from ._lib import PeaDesc

SPECIALISED_TRAIN_D = (4, 8, 16, 32, 64)  # every other width trains through the runtime-D backward (REPLICATE border excepted)


class AffinitySpec(object):
    """Everything about one call that is not a tensor."""

    def __init__(self, ndim, offsets, lam, border, norm, eps=1e-12, relu=False, act=0):
        self.ndim = int(ndim)
        self.offsets = [tuple([0] * (3 - len(o)) + [int(v) for v in o]) for o in offsets]
        self.K = len(self.offsets)
        if not 1 <= self.K <= _lib.PEA_MAX_K:
            raise ValueError("number of offsets must be in 1..%d, got %d" % (_lib.PEA_MAX_K, self.K))
        self.lam = [1.0] * self.K if lam is None else [float(v) for v in lam]
        if len(self.lam) != self.K:
            raise ValueError("lambda list must have one entry per offset")
        self.border, self.norm, self.eps = border, norm, float(eps)
        self.act = int(act)  # _lib.FLAG_* activation bits of the affs output (the loss always uses the raw cosine)
        self.relu = bool(relu) or bool(self.act & _lib.FLAG_RELU_AFFS)

    @property
    def relu(self):
        return bool(self.act & _lib.FLAG_RELU_AFFS)

    @relu.setter
    def relu(self, on):
        self.act = (self.act | _lib.FLAG_RELU_AFFS) if on else (self.act & ~_lib.FLAG_RELU_AFFS)


def _spatial(e, ndim):
    if e.dim() != ndim + 2:
        raise ValueError("embedding must be %s, got shape %s" % ("[B,D,H,W]" if ndim == 2 else "[B,D,Z,Y,X]", tuple(e.shape)))
    sp = list(e.shape[2:])
    return [1] * (3 - len(sp)) + sp


def _require_gpu(t, name):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise RuntimeError("%s is on %s: the affinity path runs on an MI355X only (no CPU fallback)" % (name, t.device))


def _embedding_arg(t, name):
    _require_gpu(t, name)
    if t.dtype not in (torch.float32, torch.float16):
        raise TypeError("%s must be float32 or float16, got %s" % (name, t.dtype))
    return t if t.is_contiguous() else t.contiguous()


def _batch_strided(t, name, dtype, shape):
    """-> (tensor, batch stride in elements).  The [K, spatial...] block of every batch item must be
    dense; the batch stride itself may be anything (channel slices of a packed tensor)."""
    _require_gpu(t, name)
    if tuple(t.shape) != tuple(shape):
        raise ValueError("%s has shape %s, expected %s" % (name, tuple(t.shape), tuple(shape)))
    if t.dtype != dtype:
        t = t.to(dtype)
    if t.shape[0] > 1 and not t[0].is_contiguous():
        t = t.contiguous()
    elif t.shape[0] == 1 and not t.is_contiguous():
        t = t.contiguous()
    return t, (t.stride(0) if t.shape[0] > 1 else 0)


_DESC_CACHE = {}


def make_desc(spec, e, tstride=0, wstride=0, mstride=0):
    """PeaDesc for `spec` on a tensor shaped like e.  Descriptors are immutable once built (the library takes them as
    const), so they are memoised: filling and validating one costs more host time than the launch it describes."""
    key = (spec.ndim, tuple(spec.offsets), tuple(spec.lam), spec.border, spec.norm, spec.eps, spec.act, tuple(e.shape),
           e.dtype == torch.float16, int(tstride), int(wstride), int(mstride))
    d = _DESC_CACHE.get(key)
    if d is not None:
        return d
    if len(_DESC_CACHE) > 512:
        _DESC_CACHE.clear()
        _CROSS_OK.clear()  # (keyed by the identity of descriptors that are about to go away)
    d = _DESC_CACHE[key] = _build_desc(spec, e, tstride, wstride, mstride)
    return d


_CROSS_OK = {}
_CROSS_LOCK = threading.Lock()


def cross_supported(d, mode):
    """pea_cross_supported(desc, mode), remembered per memoised descriptor AND current device (the host-side plan behind it costs a few
    microseconds; the march kernels' answer depends on the device's CU count, and nn.DataParallel replicas call this from one thread
    per device: round-4 advice)"""
    key = (id(d), mode, torch.cuda.current_device())
    hit = _CROSS_OK.get(key)
    if hit is None or hit[1] is not d:  # (the entry keeps its descriptor alive: an id cannot be reused while it is a key)
        r = bool(_lib.lib().pea_cross_supported(ctypes.byref(d), mode))
        with _CROSS_LOCK:
            if len(_CROSS_OK) > 2048:
                _CROSS_OK.clear()
            _CROSS_OK[key] = (r, d)
        return r
    return hit[0]


_lib._RELOAD_HOOKS.append(_CROSS_OK.clear)  # pea_cross_supported depends on the PEA_* switches


def _build_desc(spec, e, tstride, wstride, mstride):
    dims = _spatial(e, spec.ndim)
    d = PeaDesc()
    d.abi, d.ndim, d.B, d.D = _lib.PEA_ABI_VERSION, spec.ndim, e.shape[0], e.shape[1]
    d.dims[:] = dims
    d.K = spec.K
    d.border, d.norm, d.eps = spec.border, spec.norm, spec.eps
    d.dtype = _lib.F16 if e.dtype == torch.float16 else _lib.F32
    d.flags = spec.act
    for i, o in enumerate(spec.offsets):
        if spec.border == _lib.BORDER_CIRCULAR:  # torch.roll is modular: fold into (-dim, dim)
            o = tuple(int(v) - dims[a] * int(int(v) / dims[a]) if dims[a] else 0 for a, v in enumerate(o))
        d.offsets[i][:] = o
        d.lam[i] = spec.lam[i]
    d.target_bstride, d.weight_bstride, d.mask_bstride = int(tstride), int(wstride), int(mstride)
    rc = _lib.lib().pea_desc_validate(ctypes.byref(d))
    if rc:
        raise ValueError("invalid affinity descriptor: %s" % _lib.lib().pea_strerror(rc).decode())
    return d


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """the current HIP stream of the current device as a void*.  torch.cuda.current_stream() builds a Stream object through
    ~35 us of Python (device-index parsing); the raw handle is one C call -- the Python path of a training step is ~160 us of
    host time against ~215 us of GPU time, so this matters for staying GPU-bound on a busy host"""
    if _RAW_STREAM is not None:
        return ctypes.c_void_p(_RAW_STREAM(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


_WS = {}
_STATE_BYTES = [0]


def workspace(dev, desc, nstates=1):
    """(tensor, bytes): the loss-state block(s) of the training forward for the CURRENT stream of `dev` (include/pea.h,
    pea_workspace_bytes): allocated and initialised once per (device, stream), then reused by every call -- each call leaves it
    ready for the next one, and calls on one stream cannot overlap.  ~16 KB per state."""
    L = _lib.lib()
    if not _STATE_BYTES[0]:
        _STATE_BYTES[0] = int(L.pea_workspace_bytes(ctypes.byref(desc)))
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    raw = _RAW_STREAM(idx) if _RAW_STREAM is not None else torch.cuda.current_stream(dev).cuda_stream
    key = (idx, raw, nstates)
    w = _WS.get(key)
    nb = _STATE_BYTES[0] * nstates
    if w is None:
        w = torch.empty(nb // 4, dtype=torch.float32, device=dev)
        _lib.check(L.pea_workspace_init(ctypes.c_void_p(w.data_ptr()), nb, ctypes.c_void_p(raw)), "pea_workspace_init")
        _WS[key] = w
    return w, nb


class _on_device(object):
    """`with torch.cuda.device(dev)` without its cost when dev is already the current device (the one-process-per-GPU case)"""
    __slots__ = ("_ctx",)

    def __init__(self, dev):
        self._ctx = None if dev.index is None or dev.index == torch.cuda.current_device() else torch.cuda.device(dev)

    def __enter__(self):
        if self._ctx is not None:
            self._ctx.__enter__()

    def __exit__(self, *a):
        if self._ctx is not None:
            return self._ctx.__exit__(*a)
        return False


def _affs_shape(e, K):
    return (e.shape[0], K) + tuple(e.shape[2:])


def affinity_infer(e, e_other, spec):
    """affs [B,K,...] f32 = pea_affinity_infer (no autograd)."""
    e = _embedding_arg(e, "embedding")
    if e_other is not None:
        e_other = _embedding_arg(e_other, "ema_embedding").to(e.dtype)
        if e_other.shape != e.shape:
            raise ValueError("ema_embedding shape %s != embedding shape %s" % (tuple(e_other.shape), tuple(e.shape)))
    with _on_device(e.device):
        d = make_desc(spec, e)
        affs = torch.empty(_affs_shape(e, spec.K), dtype=torch.float32, device=e.device)
        _lib.check(_lib.lib().pea_affinity_infer(ctypes.byref(d), _ptr(e.detach()), _ptr(None if e_other is None else e_other.detach()),
                                                 _ptr(affs), _stream()), "pea_affinity_infer")
    return affs


class FusedAffinityMSE(torch.autograd.Function):
    """loss, affs, per_offset_losses = f(e, e_other, target, weight, mask): the WeightedMSE criterion fused into the affinity
    forward.  One forward launch (saving g = d loss / d affs and, for the self loss, the 1 / norm plane of e) and one backward
    launch.  (Round 1 also had an opt-in one-launch step on this tensor path, PEA_FUSED=1; it sampled target / weight / mask at
    p and at p - o with one-dword loads, only ever tied the two launches and is gone -- the one-launch step that pays is the
    labels-in one, LabelsAffinityMSE below.)"""

    @staticmethod
    def forward(ctx, e, e_other, target, weight, mask, spec):
        ctx.set_materialize_grads(False)  # no 95 MB zero tensors for the gradients of the non-differentiable outputs
        e_c = _embedding_arg(e, "embedding")
        o_c = None
        if e_other is not None:
            o_c = _embedding_arg(e_other, "ema_embedding").to(e_c.dtype)
            if o_c.shape != e_c.shape:
                raise ValueError("ema_embedding shape %s != embedding shape %s" % (tuple(o_c.shape), tuple(e_c.shape)))
        kshape = _affs_shape(e_c, spec.K)
        target, ts = _batch_strided(target, "target", torch.float32, kshape)
        weight, ws = _batch_strided(weight, "weightmap", torch.float32, kshape)
        ms = 0
        if mask is not None:
            if mask.dtype == torch.bool:
                mask = mask.view(torch.uint8)
            mask, ms = _batch_strided(mask, "mask", torch.uint8, kshape)
        for t in (o_c, target, weight, mask):
            if t is not None and t.device != e_c.device:
                raise RuntimeError("all operands must live on %s" % e_c.device)
        want_e = e.requires_grad
        want_o = e_other is not None and e_other.requires_grad
        with _on_device(e_c.device):
            d = make_desc(spec, e_c, ts, ws, ms)
            L = _lib.lib()
            affs = torch.empty(kshape, dtype=torch.float32, device=e_c.device)
            loss_vec = torch.empty(1 + spec.K, dtype=torch.float32, device=e_c.device)
            work, wsb = workspace(e_c.device, d)
            # g = d loss / d affs is all the backward needs besides the embeddings; skip it when nothing trains
            g = torch.empty(kshape, dtype=torch.float32, device=e_c.device) if (want_e or want_o) else None
            # 1 / norm of e, 4 bytes per pixel: what the cross backward (self loss, axis-aligned stencil) stages next to e;
            # with a detached second operand two planes (e, e_other) where the role-A cross kernels cover the shape
            inv = None
            if want_e and o_c is None and cross_supported(d, 1):
                inv = torch.empty((e_c.shape[0],) + tuple(e_c.shape[2:]), dtype=torch.float32, device=e_c.device)
            elif want_e and o_c is not None and not want_o and cross_supported(d, 2):
                inv = torch.empty((2, e_c.shape[0]) + tuple(e_c.shape[2:]), dtype=torch.float32, device=e_c.device)
            _lib.check(L.pea_affinity_fwd_ex(ctypes.byref(d), _ptr(e_c), _ptr(o_c), _ptr(target), _ptr(weight), _ptr(mask),
                                             _ptr(affs), _ptr(g), _ptr(inv), _ptr(loss_vec), _ptr(work), wsb, _stream()),
                       "pea_affinity_fwd_ex")
        ctx.spec, ctx.desc = spec, d
        ctx.has_other = o_c is not None
        # the raw cosine map is an input of the projection-first backward (pea_affinity_bwd_ex2): saved like an input, so autograd's
        # version counter catches an in-place edit of the returned map (affs.relu_()) between forward and backward
        # -- and only where the backward reads it (pea_cross_supported mode 3: the projection-first kernels at D > 16, the z-march
        # backward of 3D volumes): elsewhere saving it would turn a harmless affs.relu_() before backward() into a RuntimeError
        # (mode 4: the same for the cross loss with a detached second operand -- D = 32 / 64)
        raw = None
        if inv is not None and spec.act == 0:
            if (o_c is None and cross_supported(d, 3)) or (o_c is not None and cross_supported(d, 4)):
                raw = affs
        ctx.save_for_backward(e_c, o_c, g, inv, raw)
        loss, per_offset = loss_vec[0], loss_vec[1:]  # views of a buffer that is not itself returned
        ctx.mark_non_differentiable(affs, per_offset)
        return loss, affs, per_offset

    @staticmethod
    def backward(ctx, dloss, _daffs, _dvec):
        e_c, o_c, g, inv, raw = ctx.saved_tensors
        want_e = ctx.needs_input_grad[0]
        want_o = ctx.has_other and ctx.needs_input_grad[1]
        if not (want_e or want_o) or dloss is None:
            return None, None, None, None, None, None
        if e_c.shape[1] not in SPECIALISED_TRAIN_D and ctx.spec.border == _lib.BORDER_REPLICATE:
            raise NotImplementedError("the replicate-border backward needs D in %s (got %d)" % (SPECIALISED_TRAIN_D, e_c.shape[1]))
        with _on_device(e_c.device):
            L = _lib.lib()
            dl = dloss.to(device=e_c.device, dtype=torch.float32).contiguous()
            de = torch.empty_like(e_c) if want_e else None
            de_o = torch.empty_like(o_c) if want_o else None
            _lib.check(L.pea_affinity_bwd_ex2(ctypes.byref(ctx.desc), _ptr(e_c), _ptr(o_c), _ptr(g), _ptr(inv), _ptr(raw), _ptr(dl),
                                              _ptr(de), _ptr(de_o), _stream()), "pea_affinity_bwd_ex2")
        return de, de_o, None, None, None, None


_ONES = {}


def backward(loss, **kw):
    """loss.backward() (scripts_cvppp/main.py:311) without its per-step fill kernel: autograd seeds the backward of a scalar with
    torch.ones_like(loss) -- a launch-bound 5 us kernel plus a kernel boundary between the loss forward and its backward, 3 % of a
    215 us step.  The seed is a constant: one cached ones-scalar per device is handed to backward() instead."""
    key = (loss.device, loss.dtype)
    one = _ONES.get(key)
    if one is None:
        one = _ONES[key] = torch.ones((), dtype=loss.dtype, device=loss.device)
    loss.backward(one, **kw)


class Graphed(object):
    """fn(*static_inputs) captured ONCE in a HIP graph; replay() runs it again on whatever the static input tensors hold now.

    Why: at small shapes the path is host-bound -- one 544 x 544 image through embedding_loss + backward is 0.12 ms of Python, ctypes and
    autograd bookkeeping around 0.05 ms of kernels (BASELINE configs[0]; the BBBC 256 x 256 training crops likewise).  Everything fn
    launches (the library's kernels through the C ABI, autograd's own kernels, allocations -- taken from the graph's private pool)
    becomes one graph launch of ~10 us of host time.  fn may contain the backward (pea.backward(loss)): gradients it leaves on leaf
    tensors are captured like any other output.

        E, T, W, M = ...                                   # static inputs: refill them with .copy_() between replays
        def step(E, T, W, M):
            E.grad = None
            loss, affs, _ = pea.embedding_loss(E, T, W, M, crit, offsets)
            pea.backward(loss)
            return loss, affs, E.grad
        g = pea.graphed(step, E, T, W, M)
        loss, affs, grad = g.replay()                      # the same tensors every time (overwritten by the next replay)

    The capture is bit-identical to the eager call (same kernels, same order; tests/test_gpu_parity.py::test_graphed_step_equals_eager).
    Constraints are those of torch.cuda.graph: static shapes and addresses, no host synchronisation inside fn (the path has none)."""

    def __init__(self, fn, *static_inputs, warmup=2):
        dev = next((t.device for t in static_inputs if isinstance(t, torch.Tensor) and t.is_cuda), None)
        if dev is None:
            raise RuntimeError("graphed() needs at least one static input on a ROCm device (no CPU fallback)")
        self.static_inputs = static_inputs
        with torch.cuda.device(dev):
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):  # warm-up on the capture stream: allocator, memoised descriptors, the stream's loss-state block
                for _ in range(max(1, warmup)):
                    fn(*static_inputs)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=side):
                self.outputs = fn(*static_inputs)

    def replay(self):
        self.graph.replay()
        return self.outputs

    __call__ = replay


def graphed(fn, *static_inputs, warmup=2):
    """Graphed(fn, *static_inputs): see there"""
    return Graphed(fn, *static_inputs, warmup=warmup)


class AffinityMap(torch.autograd.Function):
    """affs = f(e, e_other) with a true vjp, for criteria other than the fused WeightedMSE."""

    @staticmethod
    def forward(ctx, e, e_other, spec):
        affs = affinity_infer(e, e_other, spec)
        e_c = _embedding_arg(e, "embedding")
        o_c = None if e_other is None else _embedding_arg(e_other, "ema_embedding").to(e_c.dtype)
        ctx.spec = spec
        ctx.has_other = o_c is not None
        ctx.save_for_backward(e_c, o_c)
        return affs

    @staticmethod
    def backward(ctx, d_affs):
        e_c, o_c = ctx.saved_tensors
        want_e = ctx.needs_input_grad[0]
        want_o = ctx.has_other and ctx.needs_input_grad[1]
        if not (want_e or want_o):
            return None, None, None
        if ctx.spec.act:
            raise NotImplementedError("an activation of the affs output is inference-only: apply it to the map yourself")
        if e_c.shape[1] not in SPECIALISED_TRAIN_D and ctx.spec.border == _lib.BORDER_REPLICATE:
            raise NotImplementedError("the replicate-border backward needs D in %s (got %d)" % (SPECIALISED_TRAIN_D, e_c.shape[1]))
        with _on_device(e_c.device):
            d = make_desc(ctx.spec, e_c)
            da = d_affs.to(torch.float32).contiguous()
            de = torch.empty_like(e_c) if want_e else None
            de_o = torch.empty_like(o_c) if want_o else None
            _lib.check(_lib.lib().pea_affinity_bwd(ctypes.byref(d), _ptr(e_c), _ptr(o_c), _ptr(da), None, _ptr(de), _ptr(de_o),
                                                   _stream()), "pea_affinity_bwd (vjp)")
        return de, de_o, None


class LossList(collections.abc.Sequence):
    """The reference's `all_loss` (a list of K Python floats filled by K `.item()` syncs,
    scripts_cvppp/loss/loss_embedding_mse.py:41) without the syncs: a read-only sequence whose floats are fetched from the
    device tensor on first access (one copy).  A Sequence, not a list subclass: C-level list operations would read the empty
    underlying storage of a lazily filled list (`all_loss + [x]`, `==`, copy, pickle); here every one of them goes through
    __getitem__ / __len__, and + / == against lists are defined to behave like the list the reference returns."""

    def __init__(self, dev_tensor):
        self.tensor = dev_tensor
        self._vals = None

    def _fill(self):
        if self._vals is None:
            self._vals = self.tensor.tolist()
        return self._vals

    def __len__(self):
        return self.tensor.numel()

    def __getitem__(self, i):
        return self._fill()[i]

    def __iter__(self):
        return iter(self._fill())

    def __repr__(self):
        return repr(self._fill())

    def __eq__(self, other):
        return list(self) == list(other) if isinstance(other, (list, tuple, LossList)) else NotImplemented

    def __add__(self, other):
        return list(self) + list(other)

    def __radd__(self, other):
        return list(other) + list(self)

    def __reduce__(self):
        return (list, (self._fill(),))  # copies / pickles as the plain list of floats


def labels_offsets_in_range(spec, e):
    """False when an offset reaches past the image (|o| >= dim): torch.roll folds such an offset modulo the dimension, which is
    what the tensor path's descriptor does, but the reference's gen_affs_ours (scripts_cvppp/utils/affinity_ours.py:17-39)
    shifts the LABEL image by the true offset -- every neighbour is then outside, mask 0, loss 0.  The labels-in kernels
    derive target / mask from the descriptor's (folded) offset, so they must not be used there: the *_from_labels wrappers
    fall back to gen_targets (true offsets) + the tensor path."""
    dims = _spatial(e, spec.ndim)
    return all(abs(o[a]) < dims[a] for o in spec.offsets for a in range(3))


ACTIVATIONS = {
    None: 0, "none": 0,
    "relu": _lib.FLAG_RELU_AFFS,                                   # F.relu(pred): scripts_cvppp/main.py:312, inference.py:193
    "mutex": _lib.FLAG_RELU_AFFS | _lib.FLAG_ONE_MINUS,            # 1 - relu(a): what seg_mutex hands to elf (utils/seg_mutex.py:4-5)
    "half": _lib.FLAG_HALF_SHIFT,                                  # (a + 1) / 2 = the L2 affinity 1 - |ehat_p - ehat_q|^2 / 4
    "half_clamp": _lib.FLAG_HALF_SHIFT | _lib.FLAG_CLAMP01,        # loss_embedding.py's embedding2affs: clamp((a + 1) / 2, 0, 1)
}


def activation_flags(activation):
    if isinstance(activation, int):
        return activation
    try:
        return ACTIVATIONS[activation]
    except KeyError:
        raise ValueError("activation must be one of %s (or FLAG_* bits), got %r" % (sorted(k for k in ACTIVATIONS if k), activation))


_RANGE_MSG = "label ids must fit int32 (and not be -2^31, the kernels' outside marker)%s: relabel the segmentation first"
_RANGE_CHECKS = collections.deque()  # (event, pinned flag) of range checks of GPU label tensors that have not been read back yet
_RANGE_LOCK = threading.Lock()       # the reference drives replicas from nn.DataParallel threads: the queue is shared


def check_label_ranges(block=True):
    """Read back the deferred range checks of _labels_int32 (GPU labels wider than int32): raises ValueError if an earlier label
    tensor held an id outside int32.  block=False only looks at checks whose copy has already arrived (no host sync); the
    labels-in entry points poll that way on every call, so a bad id surfaces one or two calls later at the latest."""
    while True:
        with _RANGE_LOCK:
            if not _RANGE_CHECKS:
                return
            ev, host = _RANGE_CHECKS[0]
            if not block and not ev.query():
                return
            _RANGE_CHECKS.popleft()
        ev.synchronize()
        if bool(host.item()):
            raise ValueError(_RANGE_MSG % " (found by the deferred range check of an earlier call)")


def _labels_int32(labels):
    """segmentation ids as the kernels take them (int32).  Wider ids are range-checked: a silent cast would turn an id >= 2^31
    negative (background for BOTH_FOREGROUND) and make ids that agree modulo 2^32 compare equal.  CPU tensors are checked at once;
    for GPU tensors the check runs on the device and its flag comes back through pinned memory without a host sync
    (check_label_ranges) -- ten blocking min / max round trips per training step was what the labels-in path was built to avoid.
    Under stream capture the check is skipped (a graph cannot raise): validate the ids where they are produced."""
    if labels.dtype in (torch.int64, torch.uint64) and labels.numel():
        l64 = labels.view(torch.int64) if labels.dtype == torch.uint64 else labels  # ids >= 2^63 come out negative: flagged too
        if not labels.is_cuda:
            lo, hi = int(l64.min()), int(l64.max())
            if lo <= -2 ** 31 or hi >= 2 ** 31:  # (-2^31 itself is the kernels' outside-the-image marker: include/pea.h)
                raise ValueError(_RANGE_MSG % (" (got %d .. %d)" % (lo, hi)))
        elif not torch.cuda.is_current_stream_capturing():
            check_label_ranges(block=False)
            top = l64 >> 31  # 0 or -1 for an id inside int32
            bad = (((top != 0) & (top != -1)) | (l64 == -2 ** 31)).any()  # (-2^31: the kernels' outside-the-image marker)
            # the flag starts out False (never garbage) and the event is recorded on the stream the copy was queued on: the current
            # stream of the LABELS' device, which need not be the current device
            host = torch.zeros((), dtype=torch.bool).pin_memory()
            with torch.cuda.device(labels.device):
                host.copy_(bad, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(labels.device))
            with _RANGE_LOCK:
                _RANGE_CHECKS.append((ev, host))
    return labels.to(torch.int32).contiguous()


LABELS_TWO_LAUNCH_MIN_PX = 1 << 18  # B * pixels from which the two-launch labels step (pea_affinity_fwd_bwd_labels_ex) pays


class LabelsStepUnsupported(NotImplementedError):
    """pea_affinity_fwd_bwd_labels has no kernel for the descriptor; the *_from_labels wrappers then generate the
    target / mask / weight tensors on the GPU (pea_gen_targets) and take the tensor path -- same results."""


class LabelsAffinityMSE(torch.autograd.Function):
    """loss, affs, per_offset_losses = f(e, e_other, labels): the training step straight from the label image
    (pea_label_weights + pea_affinity_fwd_bwd_labels): target / mask / weight are evaluated inside the kernel and
    d loss / d e comes out of the same launch (for grad_output = 1; backward() rescales it in place).
    flags: _lib.TGT_* (2D reference path: PADDING | MASK_INSIDE; 3D: BOTH_FOREGROUND)."""

    @staticmethod
    def forward(ctx, e, e_other, labels, spec, flags, need_affs=True):
        ctx.set_materialize_grads(False)
        e_c = _embedding_arg(e, "embedding")
        o_c = None
        if e_other is not None:
            o_c = _embedding_arg(e_other, "ema_embedding").to(e_c.dtype)
            if o_c.shape != e_c.shape:
                raise ValueError("ema_embedding shape %s != embedding shape %s" % (tuple(o_c.shape), tuple(e_c.shape)))
        _require_gpu(labels, "labels")
        if labels.dtype.is_floating_point or tuple(labels.shape) != (e_c.shape[0],) + tuple(e_c.shape[2:]):
            raise ValueError("labels must be an integer tensor of shape %s" % ((e_c.shape[0],) + tuple(e_c.shape[2:]),))
        lab = _labels_int32(labels)
        if not labels_offsets_in_range(spec, e_c):
            raise LabelsStepUnsupported("an offset is as long as the image: the labels-in kernels would fold it (torch.roll) where "
                                        "gen_affs_ours masks it out; use gen_targets + the tensor API")
        kshape = _affs_shape(e_c, spec.K)
        with _on_device(e_c.device):
            d = make_desc(spec, e_c)
            L = _lib.lib()
            affs = torch.empty(kshape if need_affs else (0,), dtype=torch.float32, device=e_c.device)
            loss_vec = torch.empty(1 + spec.K, dtype=torch.float32, device=e_c.device)
            wtab = torch.empty(e_c.shape[0] * spec.K * 2, dtype=torch.float32, device=e_c.device)
            cb = L.pea_targets_workspace_bytes(ctypes.byref(d))
            counts = torch.empty(max(cb, 4) // 4, dtype=torch.int32, device=e_c.device)
            _lib.check(L.pea_label_weights(ctypes.byref(d), _ptr(lab), flags, _ptr(wtab), _ptr(counts), cb, _stream()), "pea_label_weights")
            work, wsb = workspace(e_c.device, d)
            de_unit = torch.empty_like(e_c)
            # large self losses on the cross kernels: two launches (labels-in forward + cross backward) through a scratch buffer
            # for g and the 1 / norm plane; small ones are launch-bound and keep the one-launch kernel
            scratch, sb = None, 0
            if o_c is None and e_c.numel() // e_c.shape[1] >= LABELS_TWO_LAUNCH_MIN_PX:
                sb = int(L.pea_labels_scratch_bytes(ctypes.byref(d)))
                if sb:
                    scratch = torch.empty(sb // 4, dtype=torch.float32, device=e_c.device)
            rc = L.pea_affinity_fwd_bwd_labels_ex(ctypes.byref(d), _ptr(e_c), _ptr(o_c), _ptr(lab), _ptr(wtab), flags,
                                                  _ptr(affs) if need_affs else None, _ptr(loss_vec), None, _ptr(de_unit), _ptr(work), wsb,
                                                  _ptr(scratch), sb, _stream())
            if rc == _lib.E_UNSUPPORTED:
                raise LabelsStepUnsupported("no labels-in kernel for this descriptor (D != 16, image smaller than a tile, "
                                            "stencil too wide): use gen_targets + the tensor API")
            _lib.check(rc, "pea_affinity_fwd_bwd_labels")
        ctx.de_unit, ctx.desc = de_unit, d
        # for a second backward over a retained graph: saved like inputs (freed after a non-retained backward, version-checked on a
        # retained one), nothing is copied
        ctx.save_for_backward(e_c, o_c, lab, wtab)
        ctx.lflags = flags
        loss, per_offset = loss_vec[0], loss_vec[1:]
        ctx.mark_non_differentiable(affs, per_offset)
        return loss, affs, per_offset

    @staticmethod
    def _unit_gradient_again(ctx):
        """the gradient for grad_output = 1 once more (the first backward handed its buffer over, scaled in place): the step is run
        again on the saved inputs -- a retained graph is the rare case, and keeping a pristine copy would cost every step 150 MB"""
        e_c, o_c, lab, wtab = ctx.saved_tensors
        flags = ctx.lflags
        d, L = ctx.desc, _lib.lib()
        with _on_device(e_c.device):
            loss_vec = torch.empty(1 + d.K, dtype=torch.float32, device=e_c.device)
            work, wsb = workspace(e_c.device, d)
            de_unit = torch.empty_like(e_c)
            _lib.check(L.pea_affinity_fwd_bwd_labels_ex(ctypes.byref(d), _ptr(e_c), _ptr(o_c), _ptr(lab), _ptr(wtab), flags, None,
                                                        _ptr(loss_vec), None, _ptr(de_unit), _ptr(work), wsb, None, 0, _stream()),
                       "pea_affinity_fwd_bwd_labels")
        return de_unit

    @staticmethod
    def backward(ctx, dloss, _daffs, _dvec):
        if not ctx.needs_input_grad[0] or dloss is None:
            return None, None, None, None, None, None
        if ctx.de_unit is None:  # a second backward over a retained graph
            ctx.de_unit = LabelsAffinityMSE._unit_gradient_again(ctx)
        de, ctx.de_unit = ctx.de_unit, None
        with _on_device(de.device):
            dl = dloss.to(device=de.device, dtype=torch.float32).contiguous()
            _lib.check(_lib.lib().pea_scale_inplace(_ptr(de), ctx.desc.dtype, de.numel(), _ptr(dl), _stream()), "pea_scale_inplace")
        return de, None, None, None, None, None
