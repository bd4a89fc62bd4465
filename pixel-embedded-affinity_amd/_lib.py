"""ctypes binding of libpea_hip.so (the C ABI declared in include/pea.h).

The library is the product: there is NO CPU fallback.  If the shared object is missing or cannot be
loaded, every op raises PeaLibraryError loudly.  `build()` compiles it in-tree with hipcc for gfx950
(cross-compiles without a GPU), so the .so travels with the repository snapshot.
"""
import ctypes
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO_PATH = os.environ.get("PEA_HIP_LIB") or os.path.join(CSRC, "libpea_hip.so")  # PEA_HIP_LIB: debugging override
HEADER = os.path.join(HERE, "..", "include", "pea.h")

PEA_ABI_VERSION = 2
PEA_MAX_K = 32
E_UNSUPPORTED = -3
BORDER_CIRCULAR, BORDER_CROP_ZERO, BORDER_REPLICATE = 0, 1, 2
F32, F16 = 0, 1
NORM_BX, NORM_CROPPED, NORM_FULL = 0, 1, 2
FLAG_RELU_AFFS, FLAG_ONE_MINUS, FLAG_HALF_SHIFT, FLAG_CLAMP01, FLAG_ACCUMULATE_DE = 1, 2, 4, 8, 16  # activation of the affs output (include/pea.h)
TGT_PADDING, TGT_BOTH_FOREGROUND, TGT_MASK_INSIDE, TGT_ACCUMULATE = 1, 2, 4, 8

EXPORTS = ("pea_version", "pea_strerror", "pea_desc_validate", "pea_workspace_bytes", "pea_workspace_init", "pea_reload_env",
           "pea_affinity_infer", "pea_affinity_fwd", "pea_affinity_bwd", "pea_affinity_fwd_ex", "pea_affinity_bwd_ex", "pea_affinity_bwd_ex2", "pea_inv_norm",
           "pea_cross_supported", "pea_affinity_bwd_dual", "pea_affinity_bwd_dual_ex", "pea_affinity_fwd_dual_ex", "pea_scale_inplace", "pea_scale_inplace_multi", "pea_weighted_sum",
           "pea_fill_border_relu", "pea_head_workspace_bytes", "pea_head_fwd", "pea_head_bwd",
           "pea_targets_workspace_bytes", "pea_gen_targets", "pea_stitch_add", "pea_stitch_finalize",
           "pea_label_weights", "pea_affinity_fwd_bwd_labels", "pea_affinity_fwd_bwd_labels_ex", "pea_labels_scratch_bytes",
           "pea_affinity_fwd_bwd_labels_dual")


class PeaLibraryError(RuntimeError):
    """libpea_hip.so is missing / unloadable, or a pea_* call returned an error."""


class PeaDesc(ctypes.Structure):
    """mirror of `struct PeaDesc` in include/pea.h"""
    _fields_ = [("abi", ctypes.c_int32), ("ndim", ctypes.c_int32), ("B", ctypes.c_int32),
                ("D", ctypes.c_int32), ("dims", ctypes.c_int32 * 3), ("K", ctypes.c_int32),
                ("border", ctypes.c_int32), ("dtype", ctypes.c_int32), ("norm", ctypes.c_int32),
                ("flags", ctypes.c_uint32), ("eps", ctypes.c_float),
                ("offsets", (ctypes.c_int32 * 3) * PEA_MAX_K), ("lam", ctypes.c_float * PEA_MAX_K),
                ("target_bstride", ctypes.c_int64), ("weight_bstride", ctypes.c_int64),
                ("mask_bstride", ctypes.c_int64)]


HIPCC_FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC"]
OBJ_DIR = os.path.join(CSRC, "build")


def sources():
    """the translation units of the library (csrc/pea_host.h lists what each holds)"""
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def build(force=False, verbose=False, jobs=None):
    """Compile csrc/*.hip -> csrc/libpea_hip.so for gfx950: one object per file, compiled in parallel, one link.
    Serialised across processes by a lock file: the ranks of a multi-GPU job all import the package at once, and if the library
    is stale each of them would otherwise write the same object files."""
    import fcntl
    os.makedirs(OBJ_DIR, exist_ok=True)
    with open(os.path.join(OBJ_DIR, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_locked(force, verbose, jobs)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force, verbose, jobs):
    srcs = sources()
    hdrs = [HEADER] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    newest_hdr = max(os.path.getmtime(h) for h in hdrs)
    if not force and os.path.exists(SO_PATH) and os.path.getmtime(SO_PATH) >= max([newest_hdr] + [os.path.getmtime(x) for x in srcs]):
        return SO_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise PeaLibraryError("hipcc not found: cannot build libpea_hip.so")
    os.makedirs(OBJ_DIR, exist_ok=True)
    objs, todo = [], []
    for src in srcs:
        obj = os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(newest_hdr, os.path.getmtime(src)):
            todo.append((src, obj))
    jobs = jobs or min(len(todo) or 1, max(1, (os.cpu_count() or 2) - 1))
    running = []

    def reap(block_until):
        while len(running) > block_until:
            for i, (p, src, obj) in enumerate(running):
                if p.poll() is not None:
                    running.pop(i)
                    if p.returncode != 0:
                        for q, _, _ in running:
                            q.kill()
                        raise PeaLibraryError("hipcc failed on %s (rc %d)" % (src, p.returncode))
                    os.replace(obj + ".tmp", obj)
                    break
            else:
                import time
                time.sleep(0.05)

    for src, obj in todo:
        cmd = [hipcc] + HIPCC_FLAGS + ["-c", "-o", obj + ".tmp", src]
        if verbose:
            print(" ".join(cmd))
        running.append((subprocess.Popen(cmd, cwd=CSRC), src, obj))
        reap(jobs - 1)
    reap(0)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO_PATH + ".tmp"] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    os.replace(SO_PATH + ".tmp", SO_PATH)
    return SO_PATH


_lib = None


def lib():
    """Load the library (after torch, so both bind to the same HIP runtime, libamdhip64.so.7)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise PeaLibraryError(
            "%s not found: the HIP extension is not built (run `python -c 'import __graft_entry__ as g; g.build()'`). "
            "There is no CPU fallback." % SO_PATH)
    import torch  # noqa: F401  (loads torch's bundled libamdhip64 first)
    try:
        L = ctypes.CDLL(SO_PATH)
    except OSError as ex:
        raise PeaLibraryError("cannot load %s: %s" % (SO_PATH, ex))
    for name in EXPORTS:
        if not hasattr(L, name):
            raise PeaLibraryError("%s does not export %s" % (SO_PATH, name))
    vp, dp = ctypes.c_void_p, ctypes.POINTER(PeaDesc)
    L.pea_version.restype = ctypes.c_int
    L.pea_strerror.restype = ctypes.c_char_p
    L.pea_strerror.argtypes = [ctypes.c_int]
    L.pea_desc_validate.restype = ctypes.c_int
    L.pea_desc_validate.argtypes = [dp]
    L.pea_workspace_bytes.restype = ctypes.c_size_t
    L.pea_workspace_bytes.argtypes = [dp]
    L.pea_workspace_init.restype = ctypes.c_int
    L.pea_workspace_init.argtypes = [vp, ctypes.c_size_t, vp]
    L.pea_reload_env.restype = None
    L.pea_reload_env.argtypes = []
    L.pea_affinity_infer.restype = ctypes.c_int
    L.pea_affinity_infer.argtypes = [dp, vp, vp, vp, vp]
    L.pea_affinity_fwd.restype = ctypes.c_int
    L.pea_affinity_fwd.argtypes = [dp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ctypes.c_size_t, vp]
    L.pea_affinity_bwd.restype = ctypes.c_int
    L.pea_affinity_bwd.argtypes = [dp, vp, vp, vp, vp, vp, vp, vp]
    L.pea_affinity_fwd_ex.restype = ctypes.c_int
    L.pea_affinity_fwd_ex.argtypes = [dp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ctypes.c_size_t, vp]
    L.pea_affinity_bwd_ex.restype = ctypes.c_int
    L.pea_affinity_bwd_ex.argtypes = [dp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.pea_affinity_bwd_ex2.restype = ctypes.c_int
    L.pea_affinity_bwd_ex2.argtypes = [dp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.pea_cross_supported.restype = ctypes.c_int
    L.pea_cross_supported.argtypes = [dp, ctypes.c_int]
    L.pea_inv_norm.restype = ctypes.c_int
    L.pea_inv_norm.argtypes = [dp, vp, vp, vp]
    L.pea_affinity_bwd_dual_ex.restype = ctypes.c_int
    L.pea_affinity_bwd_dual_ex.argtypes = [dp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.pea_affinity_fwd_dual_ex.restype = ctypes.c_int
    L.pea_affinity_fwd_dual_ex.argtypes = [dp, dp] + [vp] * 14 + [ctypes.c_size_t, vp]
    L.pea_affinity_bwd_dual.restype = ctypes.c_int
    L.pea_affinity_bwd_dual.argtypes = [dp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.pea_scale_inplace.restype = ctypes.c_int
    L.pea_scale_inplace.argtypes = [vp, ctypes.c_int, ctypes.c_size_t, vp, vp]
    L.pea_targets_workspace_bytes.restype = ctypes.c_size_t
    L.pea_targets_workspace_bytes.argtypes = [dp]
    L.pea_gen_targets.restype = ctypes.c_int
    L.pea_gen_targets.argtypes = [dp, vp, ctypes.c_uint, vp, vp, vp, vp, ctypes.c_size_t, vp]
    L.pea_scale_inplace_multi.restype = ctypes.c_int
    L.pea_scale_inplace_multi.argtypes = [ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_size_t), ctypes.c_int, ctypes.c_int, vp, vp]
    L.pea_weighted_sum.restype = ctypes.c_int
    L.pea_weighted_sum.argtypes = [vp, ctypes.c_int, vp, ctypes.c_int, vp, vp]
    L.pea_label_weights.restype = ctypes.c_int
    L.pea_label_weights.argtypes = [dp, vp, ctypes.c_uint, vp, vp, ctypes.c_size_t, vp]
    L.pea_affinity_fwd_bwd_labels.restype = ctypes.c_int
    L.pea_affinity_fwd_bwd_labels.argtypes = [dp, vp, vp, vp, vp, ctypes.c_uint, vp, vp, vp, vp, vp, ctypes.c_size_t, vp]
    L.pea_affinity_fwd_bwd_labels_ex.restype = ctypes.c_int
    L.pea_affinity_fwd_bwd_labels_ex.argtypes = [dp, vp, vp, vp, vp, ctypes.c_uint, vp, vp, vp, vp, vp, ctypes.c_size_t, vp, ctypes.c_size_t, vp]
    L.pea_labels_scratch_bytes.restype = ctypes.c_size_t
    L.pea_labels_scratch_bytes.argtypes = [dp]
    L.pea_affinity_fwd_bwd_labels_dual.restype = ctypes.c_int
    L.pea_affinity_fwd_bwd_labels_dual.argtypes = [dp, dp, vp, vp, vp, vp, ctypes.c_uint, vp, vp, vp, vp, vp, vp, vp, ctypes.c_size_t, vp]
    L.pea_stitch_add.restype = ctypes.c_int
    L.pea_stitch_add.argtypes = [vp, vp, vp, vp] + [ctypes.c_int] * 10 + [vp]
    L.pea_stitch_finalize.restype = ctypes.c_int
    L.pea_stitch_finalize.argtypes = [vp, vp, ctypes.c_int, ctypes.c_size_t, vp]
    L.pea_fill_border_relu.restype = ctypes.c_int
    L.pea_fill_border_relu.argtypes = [vp] + [ctypes.c_int] * 7 + [vp]
    L.pea_head_workspace_bytes.restype = ctypes.c_size_t
    L.pea_head_workspace_bytes.argtypes = [ctypes.c_int, ctypes.c_int]
    L.pea_head_fwd.restype = ctypes.c_int
    L.pea_head_fwd.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_size_t, vp, vp, vp, vp, vp]
    L.pea_head_bwd.restype = ctypes.c_int
    L.pea_head_bwd.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_size_t, vp, vp, vp, vp, vp, vp, vp, ctypes.c_size_t, vp]
    if L.pea_version() != PEA_ABI_VERSION:
        raise PeaLibraryError("ABI mismatch: library %d, binding %d" % (L.pea_version(), PEA_ABI_VERSION))
    _lib = L
    return L


_RELOAD_HOOKS = []


def reload_env():
    """make the loaded library re-read the PEA_* switches (it reads them once) and drop what the Python layer remembered of them"""
    if _lib is not None:
        _lib.pea_reload_env()
    for hook in _RELOAD_HOOKS:
        hook()


def set_switch(name, value):
    """set (value) or clear (None) a PEA_* environment switch, then reload_env()"""
    if value is None:
        os.environ.pop(name, None)
    else:
        os.environ[name] = str(value)
    reload_env()


def check(rc, what):
    if rc != 0:
        raise PeaLibraryError("%s failed: rc=%d (%s)" % (what, rc, lib().pea_strerror(rc).decode()))
