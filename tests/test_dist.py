"""CPU, world_size 2 over gloo: the batch-sharding identity the multi-GPU path relies on.  Each rank computes the
op on its shard with the LOCAL normaliser (the CPU oracle stands in for the HIP kernels, which need a GPU);
DDP-style averaging over ranks must reproduce the full-batch loss and the full-batch backbone gradient."""
import importlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import __graft_entry__ as ge


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ge.load_package()
        orc = ge.load_oracle()
        synth = importlib.import_module(ge.PKG_NAME + ".utils.synth")
        sh = importlib.import_module(ge.PKG_NAME + ".utils.shard")
        offsets = orc.multi_offset([1, 3, 5, 9], 4)
        B, D, H, W = 4, 16, 24, 40
        e, t, w, m = synth.synth_inputs_2d(B, D, H, W, offsets, 31)
        theta = 0.7  # one shared "backbone" parameter: embedding = theta * x
        lo, hi = sh.shard_range(B, rank, world)
        xs, ts, ws, ms = e[lo:hi], t[lo:hi], w[lo:hi], m[lo:hi]
        d = orc.desc_2d(theta * xs, offsets)
        _, loss = orc.c_fwd(d, theta * xs, None, ts, ws, ms)
        de, _ = orc.c_bwd(d, theta * xs, None, ts, ws, ms)
        dtheta = torch.tensor([float((de.astype(np.float64) * xs).sum())], dtype=torch.float64)
        logged = torch.tensor(loss, dtype=torch.float64)  # loss and the K per-offset losses
        sh.allreduce_mean_([dtheta, logged])
        if rank == 0:
            df = orc.desc_2d(theta * e, offsets)
            _, loss_full = orc.c_fwd(df, theta * e, None, t, w, m)
            de_full, _ = orc.c_bwd(df, theta * e, None, t, w, m)
            dtheta_full = float((de_full.astype(np.float64) * e).sum())
            out.put((logged.numpy(), loss_full, float(dtheta), dtheta_full))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_sharded_loss_and_gradient_match_full_batch_gloo_world2():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = out.get(timeout=150)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    logged, loss_full, dtheta, dtheta_full = res
    np.testing.assert_allclose(logged, loss_full, rtol=1e-9)
    assert abs(dtheta - dtheta_full) <= 1e-6 * abs(dtheta_full)


def test_shard_range_contract(pkg):
    sh = importlib.import_module(ge.PKG_NAME + ".utils.shard")
    assert [sh.shard_range(8, r, 4) for r in range(4)] == [(0, 2), (2, 4), (4, 6), (6, 8)]
    with pytest.raises(ValueError, match="cannot be equally divided"):
        sh.shard_range(6, 0, 4)
    t = torch.arange(8)
    assert sh.shard(t, 1, 2).tolist() == [4, 5, 6, 7]
    assert sh.allreduce_mean_(torch.ones(2)).tolist() == [1.0, 1.0]  # single process: no-op
