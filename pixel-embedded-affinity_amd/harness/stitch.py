"""3D inference stitcher on the device: Provider_valid.reset_output / get_weight / add_vol / get_results of
scripts_ac3ac4/data/provider_valid.py:291-349 (model_type 'superhuman') without the per-window D2H copy and numpy
accumulation (scripts_ac3ac4/inference.py:166, main.py:302).  The Gaussian blend weights are computed exactly as the
reference does (numpy, float32 linspace / meshgrid) once on the host; everything per window runs in pea_stitch_add.
"""
import ctypes

import numpy as np
import torch

from .. import _lib


def get_weight(out_size, sigma=0.2, mu=0.0):
    """provider_valid.py:305-318 (num_z >= 18 branch): [1, oz, oy, ox] float32"""
    zz, yy, xx = np.meshgrid(np.linspace(-1, 1, out_size[0], dtype=np.float32),
                             np.linspace(-1, 1, out_size[1], dtype=np.float32),
                             np.linspace(-1, 1, out_size[2], dtype=np.float32), indexing='ij')
    dd = np.sqrt(zz * zz + yy * yy + xx * xx)
    weight = 1e-6 + np.exp(-((dd - mu) ** 2 / (2.0 * sigma ** 2)))
    return weight[np.newaxis, ...]


class VolumeStitcher(object):
    """out_affs [C, Z, Y, X] / weight_map [1, Z, Y, X] on the GPU; windows of out_size = (oz, oy, ox)."""

    def __init__(self, channels, vol_shape, out_size, device, sigma=0.2, mu=0.0):
        self.C, self.shape, self.out_size = int(channels), tuple(int(v) for v in vol_shape), tuple(int(v) for v in out_size)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("VolumeStitcher runs on an MI355X only (no CPU fallback)")
        self.weight_vol = torch.from_numpy(np.ascontiguousarray(get_weight(self.out_size, sigma, mu), dtype=np.float32)).to(self.device)
        self.reset_output()

    def reset_output(self):
        """provider_valid.py:291-303"""
        self.out_affs = torch.zeros((self.C,) + self.shape, dtype=torch.float32, device=self.device)
        self.weight_map = torch.zeros((1,) + self.shape, dtype=torch.float32, device=self.device)

    def add_vol(self, affs_vol, pos):
        """provider_valid.py:320-331.  affs_vol [C, oz, oy, ox] (f32, on the GPU); pos = (z0, y0, x0) start of the window
        in the volume.  (The reference slices the 2nd spatial dim with `fromx` and the 3rd with `fromy`, :326-331; its
        windows are square in y/x, so pass pos = (fromz, fromx, fromy) to reproduce it literally.)"""
        if not affs_vol.is_cuda or affs_vol.dtype != torch.float32:
            raise ValueError("affs_vol must be a float32 tensor on the GPU")
        v = affs_vol.reshape((self.C,) + self.out_size).contiguous()
        Z, Y, X = self.shape
        oz, oy, ox = self.out_size
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().pea_stitch_add(p(self.out_affs), p(self.weight_map), p(v), p(self.weight_vol), self.C, Z, Y, X,
                                                 oz, oy, ox, int(pos[0]), int(pos[1]), int(pos[2]),
                                                 ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "pea_stitch_add")

    def get_results(self, valid_padding):
        """provider_valid.py:337-349: out / weight_map, cropped by valid_padding (a view of the device tensor)"""
        Z, Y, X = self.shape
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().pea_stitch_finalize(p(self.out_affs), p(self.weight_map), self.C, Z * Y * X,
                                                      ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "pea_stitch_finalize")
        vz, vy, vx = (int(v) for v in valid_padding)
        out = self.out_affs
        if vz:
            out = out[:, vz:-vz]
        return out[:, :, vy:-vy, vx:-vx] if (vy and vx) else out
