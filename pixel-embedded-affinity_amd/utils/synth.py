"""Deterministic, RNG-free synthetic inputs for the affinity path (bench.py, smoke(), tests).

Everything is a closed-form integer hash of the element index, evaluated in exact integer /
float64 arithmetic with no transcendental functions, so the same (shape, seed) gives bit-identical
arrays on any host.  The shapes and value distributions follow SURVEY.md section 8d:

  e       ~ approx N(0,1)                      (sum of 4 uniforms, variance-normalised)
  labels  = blocky instance map, background 0 + n instances
  target, mask = what gen_affs_ours(labels, offsets, ignore=False, padding=True) produces
            (scripts_cvppp/utils/affinity_ours.py:17-39: t=1 iff label(p)==label(p+o); a neighbour
            outside the image gives t=1, mask=0)
  weight  = per-channel class balance as weight_binary_ratio does
            (scripts_cvppp/data/data_segmentation.py:205-228)
"""
import numpy as np

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_G = np.uint64(0x9E3779B97F4A7C15)


def hash_u64(idx, seed):
    """splitmix64 finaliser of (idx + seed * golden); idx: uint64 array."""
    with np.errstate(over="ignore"):
        z = idx.astype(np.uint64) + np.uint64(seed) * _G
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def hash_uniform(idx, seed):
    """uniform in [0,1) as float64, exact (53-bit mantissa from the hash)."""
    return (hash_u64(idx, seed) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def synth_embedding(shape, seed):
    n = int(np.prod(shape))
    idx = np.arange(n, dtype=np.uint64)
    s = np.zeros(n, np.float64)
    for k in range(4):
        s += hash_uniform(idx, seed * 4 + k)
    return ((s - 2.0) * np.sqrt(3.0)).astype(np.float32).reshape(shape)


def synth_labels(B, dims, seed, cell=16, n=15):
    """[B, Z, Y, X] int32 blocky instance labels; 0 = background."""
    Z, Y, X = dims
    cz = max(1, min(cell, 4))
    gz, gy, gx = -(-Z // cz), -(-Y // cell), -(-X // cell)
    idx = np.arange(B * gz * gy * gx, dtype=np.uint64)
    coarse = (hash_u64(idx, seed + 7919) % np.uint64(n + 1)).astype(np.int32).reshape(B, gz, gy, gx)
    lab = coarse.repeat(cz, axis=1).repeat(cell, axis=2).repeat(cell, axis=3)
    return np.ascontiguousarray(lab[:, :Z, :Y, :X])


def affinity_targets(labels, offsets, padding=True, both_foreground=False):
    """labels [B,Z,Y,X] -> target f32 [B,K,Z,Y,X], mask u8 [B,K,Z,Y,X] (mask = neighbour inside the image)."""
    B, Z, Y, X = labels.shape
    K = len(offsets)
    t = np.zeros((B, K, Z, Y, X), np.float32)
    m = np.zeros((B, K, Z, Y, X), np.uint8)
    for i, (dz, dy, dx) in enumerate(offsets):
        z0, z1 = max(0, -dz), min(Z, Z - dz)
        y0, y1 = max(0, -dy), min(Y, Y - dy)
        x0, x1 = max(0, -dx), min(X, X - dx)
        t[:, i] = 1.0 if padding else 0.0
        if z0 < z1 and y0 < y1 and x0 < x1:
            a = labels[:, z0:z1, y0:y1, x0:x1]
            b = labels[:, z0 + dz:z1 + dz, y0 + dy:y1 + dy, x0 + dx:x1 + dx]
            eq = a == b
            if both_foreground:
                eq &= (a > 0) & (b > 0)
            t[:, i, z0:z1, y0:y1, x0:x1] = eq
            m[:, i, z0:z1, y0:y1, x0:x1] = 1
    return t, m


def class_balance_weights(t):
    """per (b, channel) class-balance weights of a binary target."""
    w = np.ones_like(t, dtype=np.float32)
    B, K = t.shape[:2]
    for b in range(B):
        for i in range(K):
            lab = t[b, i] != 0
            if lab.all() or not lab.any():
                continue
            f = float(np.clip(lab.mean(), 5e-2, 0.99))
            if f > 0.5:
                w[b, i] = np.where(lab, 1.0, f / (1 - f))
            else:
                w[b, i] = np.where(lab, (1 - f) / f, 1.0)
    return w


def _offsets3(offsets):
    return [([0] * (3 - len(o)) + [int(v) for v in o]) for o in offsets]


def synth_inputs_2d(B, D, H, W, offsets, seed):
    """-> e [B,D,H,W] f32, target/weight [B,K,H,W] f32, mask [B,K,H,W] u8"""
    e = synth_embedding((B, D, H, W), seed)
    lab = synth_labels(B, (1, H, W), seed)
    t, m = affinity_targets(lab, _offsets3(offsets), padding=True)
    w = class_balance_weights(t)
    sq = lambda a: np.ascontiguousarray(a[:, :, 0])
    return e, sq(t), sq(w), sq(m)


def synth_inputs_3d(B, D, Z, Y, X, offsets, seed):
    """-> e [B,D,Z,Y,X], target/weight [B,K,Z,Y,X] (3D path has no mask; seg_to_aff semantics:
    t=1 iff equal AND both foreground, scripts_ac3ac4/data/data_affinity.py:53-102)"""
    e = synth_embedding((B, D, Z, Y, X), seed)
    lab = synth_labels(B, (Z, Y, X), seed)
    t, _ = affinity_targets(lab, _offsets3(offsets), padding=False, both_foreground=True)
    w = class_balance_weights(t)
    return e, t, w
