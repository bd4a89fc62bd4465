"""Hand-off of the affinity maps to the reference's CPU post-processing: what inference.py does between
the network and elf / waterz / h5py, without the per-image device -> pageable-host copy and the numpy statements in between.

    scripts_cvppp/inference.py:193-202   pred = F.relu(pred); output_affs = squeeze(pred.cpu().numpy()); seg_mutex(output_affs, ...)
    scripts_cvppp/utils/seg_mutex.py:4-5 mutex_watershed(1.0 - affs, offsets, strides, mask=...)
    scripts_cvppp/inference.py:295-299   affs.hdf: dataset 'main' = float32 [N, K, H, W], gzip

AffsCollector keeps ONE pinned host array [N, K, H, W] (the layout of affs.hdf) and copies every image's map into its slot with
an asynchronous D2H on the caller's stream; `mutex_input(i)` returns 1 - affs for elf as a view-free numpy array.  The maps
are produced already clamped: embedding2affs(..., activation='relu') writes max(a, 0) with the kernel's own store."""
import numpy as np
import torch


class AffsCollector(object):
    def __init__(self, n_images, K, H, W):
        self.host = torch.empty((n_images, K, H, W), dtype=torch.float32).pin_memory() if torch.cuda.is_available() else \
            torch.empty((n_images, K, H, W), dtype=torch.float32)
        self.n = 0
        self._events = []

    def add(self, affs):
        """affs: [1, K, H, W] or [K, H, W] float32 on the GPU (relu already applied); returns the image's index"""
        a = affs.reshape(self.host.shape[1:])
        i = self.n
        self.host[i].copy_(a, non_blocking=True)
        if a.is_cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(a.device))
            self._events.append(ev)
        self.n += 1
        return i

    def wait(self):
        for ev in self._events:
            ev.synchronize()
        self._events = []

    def numpy(self):
        """float32 [N, K, H, W], C-contiguous: the array inference.py:296-298 writes to affs.hdf"""
        self.wait()
        return self.host[:self.n].numpy()

    def mutex_input(self, i):
        """1 - affs of image i, float32 [K, H, W]: the first argument of elf's mutex_watershed (seg_mutex.py:5)"""
        self.wait()
        return 1.0 - self.host[i].numpy()

    def save_hdf(self, path):
        """affs.hdf as scripts_cvppp/inference.py:295-299 writes it (needs h5py, which this package does not depend on)"""
        import h5py
        with h5py.File(path, "w") as f:
            a = self.numpy()
            f.create_dataset("main", data=a, dtype=a.dtype, compression="gzip")
