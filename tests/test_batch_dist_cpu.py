"""The N>1 path of the batch stage on CPU: two processes (gloo, 127.0.0.1), each linearises the constraints of
its keyframe shard (with the CPU oracle standing in for the HIP kernel), one all-reduce sums the block-banded
[H|g|cost] buffers -- the same driver logic (glio_amd.batch.shard_range / hg layout) the GPU path uses."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from glio_amd import batch


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, K, band, per_kf, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pyoracle as po
    gt, init = batch.make_poses(K, seed=4)
    lo, hi = batch.shard_range(K, rank, world)
    ci, cj, cp, nc, score = batch.make_constraints(gt, lo, hi, per_kf, band, seed=4)
    Hb, g, cost = po.batch_linearize(K, band, np.ascontiguousarray(init), ci, cj, cp.numpy(), nc.numpy(), score.numpy())
    Hg = torch.from_numpy(np.concatenate([Hb.ravel(), g.ravel(), [cost]]))
    assert Hg.numel() == batch.hg_size(K, band)
    dist.all_reduce(Hg, op=dist.ReduceOp.SUM)
    if rank == 0:
        np.save(out_path, Hg.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_ranges_partition_the_keyframes():
    for K in (7, 2000, 2001):
        for world in (1, 2, 3, 8):
            r = [batch.shard_range(K, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == K and all(a[1] == b[0] for a, b in zip(r, r[1:]))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_allreduce_equals_single_rank(tmp_path):
    from oracle import pyoracle as po
    K, band, per_kf, world = 20, 4, 60, 2
    out = str(tmp_path / "hg.npy")
    mp.spawn(_worker, args=(world, _free_port(), K, band, per_kf, out), nprocs=world, join=True)
    reduced = np.load(out)
    gt, init = batch.make_poses(K, seed=4)
    parts = [batch.make_constraints(gt, *batch.shard_range(K, r, world), per_kf, band, seed=4) for r in range(world)]
    ci = np.concatenate([p[0] for p in parts]); cj = np.concatenate([p[1] for p in parts])
    cp = np.concatenate([p[2].numpy() for p in parts]); nc = np.concatenate([p[3].numpy() for p in parts]); sc = np.concatenate([p[4].numpy() for p in parts])
    Hb, g, cost = po.batch_linearize(K, band, np.ascontiguousarray(init), ci, cj, cp, nc, sc)
    full = np.concatenate([Hb.ravel(), g.ravel(), [cost]])
    assert np.linalg.norm(reduced - full) <= 1e-12 * np.linalg.norm(full)
