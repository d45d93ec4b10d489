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
    """whole super-blocks of the block cyclic reduction (6 keyframes; 12 for bands > 6), spread as evenly as whole blocks allow;
    the Python rule equals the library's (glio_batch_shard_range is host-only code: no GPU needed)"""
    import ctypes as C
    from glio_amd import capi
    lib = capi.load()
    for band in (4, 6, 12):
        sbk = 6 if band <= 6 else 12
        for K in (50, 2000, 2001):
            for world in (1, 2, 3, 8):
                if (K + sbk - 1) // sbk < world:
                    continue
                r = [batch.shard_range(K, k, world, band) for k in range(world)]
                assert r[0][0] == 0 and r[-1][1] == K and all(a[1] == b[0] for a, b in zip(r, r[1:]))
                sizes = [b - a for a, b in r]
                assert all(sz % sbk == 0 for sz in sizes[:-1]) and max(sizes) - min(sizes) <= 2 * sbk
                for k in range(world):
                    lo, hi = C.c_int32(), C.c_int32()
                    assert lib.glio_batch_shard_range(K, band, k, world, C.byref(lo), C.byref(hi)) == 0
                    assert (lo.value, hi.value) == r[k]


def test_thread_ranks_harness_sums_in_rank_order():
    """batch.ThreadRanks (the in-process stand-in for torch.distributed that the one-GPU tests of the sharded solve use)"""
    world = 3
    tr = batch.ThreadRanks(world)

    def work(r, d):
        t = torch.full((5,), float(r + 1), dtype=torch.float64)
        d.all_reduce(t)
        d.all_reduce(t)
        return t

    out = tr.run(work)
    assert all(torch.equal(o, torch.full((5,), 18.0, dtype=torch.float64)) for o in out) and tr.calls == [2, 2, 2]


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


# ---- association + linearisation of the batch stage, sharded by source keyframe (oracle standing in for the HIP kernels)
def _frames(K=6, pts=500):
    from glio_amd import synth
    win = synth.make_window(W=K, pts_per_scan=pts, seed=synth.SEED_BASE + 61, perturb=(0.03, 0.2, 0.0), scan_radius=12.0, map_density=1.0)
    tlb = np.array(win.opts.t_lb, np.float32)
    scans = []
    for s in range(K):
        c = win.scans[s].copy(); c[:, :3] -= tlb
        scans.append(np.ascontiguousarray(c))
    return scans, np.ascontiguousarray(np.c_[win.init.trans, win.init.quat])


def _associate_and_linearize(po, scans, poses, ci, cj, K, band):
    cps, ncs, scs, cis, cjs = [], [], [], [], []
    for a, b in zip(ci, cj):
        cp, nc, sc, _ = po.associate_pair(scans[a], poses[a], scans[b], poses[b])
        cps.append(cp); ncs.append(nc); scs.append(sc); cis.append(np.full(len(sc), a, np.int32)); cjs.append(np.full(len(sc), b, np.int32))
    cat = lambda xs, shape: np.concatenate(xs) if xs else np.zeros(shape)
    Hb, g, cost = po.batch_linearize(K, band, poses, cat(cis, 0).astype(np.int32), cat(cjs, 0).astype(np.int32),
                                     cat(cps, (0, 4)).astype(np.float32), cat(ncs, (0, 6)), cat(scs, 0))
    return np.concatenate([Hb.ravel(), g.ravel(), [cost]])


def _assoc_worker(rank, world, port, K, rng, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pyoracle as po
    scans, poses = _frames(K)
    ci, cj = batch.pair_list(K, rng)
    first, last = batch.pair_shard(ci, K, rank, world, 2 * rng)
    Hg = torch.from_numpy(_associate_and_linearize(po, scans, poses, ci[first:last], cj[first:last], K, 2 * rng))
    dist.all_reduce(Hg, op=dist.ReduceOp.SUM)
    if rank == 0:
        np.save(out_path, Hg.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_pair_shards_partition_the_pair_list():
    for K, rng in ((6, 1), (20, 3), (33, 6)):
        ci, cj = batch.pair_list(K, rng)
        for world in (1, 2, 3, 8):
            band = 2 * rng          # the end windows of pair_list reach 2 * search_range: the band of the stage that takes these pairs
            cuts = [batch.pair_shard(ci, K, r, world, band) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == len(ci) and all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
            for r, (f, l) in enumerate(cuts):
                lo, hi = batch.shard_range(K, r, world, band)       # (band 12 cuts on 12-keyframe super-blocks, not the default 6)
                assert np.all((ci[f:l] >= lo) & (ci[f:l] < hi))


def test_two_rank_association_plus_linearisation_equals_single_rank(tmp_path):
    from oracle import pyoracle as po
    K, rng, world = 6, 1, 2
    out = str(tmp_path / "hg_assoc.npy")
    mp.spawn(_assoc_worker, args=(world, _free_port(), K, rng, out), nprocs=world, join=True)
    reduced = np.load(out)
    scans, poses = _frames(K)
    ci, cj = batch.pair_list(K, rng)
    full = _associate_and_linearize(po, scans, poses, ci, cj, K, 2 * rng)
    assert full[-1] > 0, "the synthetic frames must produce constraints"
    assert np.abs(reduced - full).max() <= 1e-9 * np.abs(full).max()


# ---- the driver itself (glio_amd.batch.ShardedBatchSolve: the object bench.py runs on the GPUs) on two gloo ranks, with a CPU
# stand-in for the HIP stage: same interface, lineariser = the oracle, step = a dense numpy solve of the band
class _OracleStage:
    def __init__(self, K, band, ci, cj, cp, nc, score):
        self.K, self.band, self.c = K, band, (ci, cj, cp, nc, score)

    def new_hg(self):
        return torch.zeros(batch.hg_size(self.K, self.band), dtype=torch.float64)

    def linearize(self, poses, Hg):
        from oracle import pyoracle as po
        ci, cj, cp, nc, score = self.c
        Hb, g, cost = po.batch_linearize(self.K, self.band, np.ascontiguousarray(poses), ci, cj, cp, nc, score)
        Hg.copy_(torch.from_numpy(np.concatenate([Hb.ravel(), g.ravel(), [cost]])))

    def step(self, Hg, lam, poses):
        from oracle import pyoracle as po
        Hb, g, cost = batch.unpack_hg(Hg.numpy(), self.K, self.band)
        H = batch.dense_from_band(Hb, self.K, self.band)
        d = np.linalg.solve(H + np.diag(lam * np.diag(H) + 1e-12), -g.ravel()).reshape(self.K, 6)
        out = poses.copy()
        out[:, :3] += d[:, :3]
        for k in range(self.K):
            out[k, 3:] = po.quat_plus(poses[k, 3:], d[k, 3:])
        return out, float(-(g.ravel() @ d.ravel() + 0.5 * d.ravel() @ H @ d.ravel()))


def _driver_worker(rank, world, port, K, band, per_kf, iters, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    gt, init = batch.make_poses(K, seed=6)
    ci, cj, cp, nc, score = batch.make_constraints(gt, 0, K, per_kf, band, seed=6)       # the global set, cut by source keyframe
    lo, hi = batch.shard_range(K, rank, world)
    a0, a1 = int(np.searchsorted(ci, lo, side="left")), int(np.searchsorted(ci, hi, side="left"))
    stage = _OracleStage(K, band, ci[a0:a1], cj[a0:a1], cp.numpy()[a0:a1], nc.numpy()[a0:a1], score.numpy()[a0:a1])
    drv = batch.ShardedBatchSolve(stage, dist if world > 1 else None)
    poses, hist = drv.solve(init, iterations=iters)
    assert drv.allreduces == (iters + 1 if world > 1 else 0)
    if rank == 0:
        np.save(out_path, np.concatenate([poses.ravel(), hist]))
    if world > 1:
        dist.barrier()
    dist.destroy_process_group()


def test_sharded_driver_two_ranks_equal_one_rank(tmp_path):
    """ShardedBatchSolve end to end (linearise my shard, one all-reduce, identical step on every rank, accept/reject) on two
    gloo ranks reproduces the one-rank solve of the same constraint set."""
    K, band, per_kf, iters = 24, 4, 80, 3
    outs = []
    for world in (1, 2):
        out = str(tmp_path / f"drv{world}.npy")
        mp.spawn(_driver_worker, args=(world, _free_port(), K, band, per_kf, iters, out), nprocs=world, join=True)
        outs.append(np.load(out))
    assert np.abs(outs[0] - outs[1]).max() <= 1e-9 * max(1.0, np.abs(outs[0]).max())
    hist = outs[0][-(iters + 1):]
    assert hist[-1] < 0.2 * hist[0]
