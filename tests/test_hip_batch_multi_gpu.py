"""The sharded batch solve across PROCESSES.
* over RCCL with one process per GPU: needs >= 2 visible devices, skipped on a one-GPU box;
* two and three processes SHARING one GPU, collectives through gloo (host copies -- RCCL refuses two ranks on one device): runs on a
  one-GPU box and exercises everything but RCCL itself -- separate address spaces, the rendezvous, the deterministic collective
  sequence of the device-resident loop (a rank that took a different decision would leave the others waiting: the test has a
  timeout), the HIP stage on every rank.
(The same sharding logic also runs on virtual ranks, tests/test_hip_batch_tr.py::test_sharded_solve_on_virtual_ranks_equals_one_rank,
and under gloo with a CPU stand-in, tests/test_batch_dist_cpu.py.)"""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _problem(K=96, band=6, per_kf=120, seed=57):
    from glio_amd import batch
    gt, init = batch.make_poses(K, seed=seed, perturb=(0.08, 0.004))
    ci, cj, cp, nc, score = batch.make_constraints(gt, 0, K, per_kf, band, seed=seed)
    odo = gt.copy(); odo[:, :3] += np.random.default_rng(seed).normal(0, 0.02, (K, 3))
    dq = batch.delta_q_pairs(odo, 3)
    dd, frame = batch.make_batch_gnss(gt, seed=seed)
    for f in dd:
        f.threshold = 10.0
    imu, sb_gt, sb0 = batch.make_batch_imu(K, seed=seed)
    return K, band, init, (ci, cj, cp.numpy(), nc.numpy(), score.numpy()), dq, dd, frame, imu, sb0


def _rank(rank, world, port, out_path, share_gpu=False):
    import datetime
    import torch
    import torch.distributed as dist
    from glio_amd import batch
    from glio_amd import ctypes_types as T
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    device = 0 if share_gpu else rank
    torch.cuda.set_device(device)
    if share_gpu:
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    else:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    K, band, init, con, dq, dd, frame, imu, sb0 = _problem()
    lo, hi = batch.shard_range(K, rank, world, band)
    own = (con[0] >= lo) & (con[0] < hi)
    st = batch.BatchStage(K, band, max(1, int(own.sum())), device=device)
    st.set_shard(rank, world)
    st.set_constraints(*[c[own] for c in con])
    st.set_small_factors(dq, dd, frame)
    st.set_imu(imu)
    poses, sb, summ = st.solve_tr(init, T.batch_tr_opts(max_iterations=12), dist, speed_bias=sb0)
    if rank == 0:
        np.savez(out_path, poses=poses, sb=sb, iterations=summ.iterations, termination=summ.termination, cost=summ.final_cost, calls=st.allreduces)
    dist.barrier()
    dist.destroy_process_group()


def _one_rank_and_compare(two):
    from glio_amd import batch
    from glio_amd import ctypes_types as T
    K, band, init, con, dq, dd, frame, imu, sb0 = _problem()
    st = batch.BatchStage(K, band, len(con[0]))
    st.set_constraints(*con); st.set_small_factors(dq, dd, frame); st.set_imu(imu)
    poses, sb, summ = st.solve_tr(init, T.batch_tr_opts(max_iterations=12), speed_bias=sb0)
    st.close()
    assert int(two["iterations"]) == summ.iterations and int(two["termination"]) == summ.termination
    assert np.isclose(float(two["cost"]), summ.final_cost, rtol=1e-9)
    assert np.abs(two["poses"] - poses).max() < 1e-9 and np.abs(two["sb"] - sb).max() < 1e-8
    assert int(two["calls"]) >= 1 + 5 * summ.iterations


def test_two_gpus_over_rccl_equal_one_gpu(tmp_path):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs (one process per GPU over RCCL)")
    import torch.multiprocessing as mp
    out = str(tmp_path / "two.npz")
    mp.spawn(_rank, args=(2, _free_port(), out), nprocs=2, join=True)
    _one_rank_and_compare(np.load(out))


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 3])
def test_processes_sharing_one_gpu_equal_one_rank(tmp_path, world):
    import torch.multiprocessing as mp
    out = str(tmp_path / "shared.npz")
    mp.spawn(_rank, args=(world, _free_port(), out, True), nprocs=world, join=True)
    _one_rank_and_compare(np.load(out))


def _rccl_hook_rank(rank, world, port, out_path):
    import torch
    import torch.distributed as dist
    from glio_amd import batch
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    st = batch.BatchStage(12, 6, 16)
    cb, calls = st._hook(dist)
    side = torch.cuda.Stream()                                   # stands for the library's stream
    buf = torch.arange(1000, dtype=torch.float64, device="cuda")
    with torch.cuda.stream(side):
        buf.mul_(2.0)                                            # work queued on that stream BEFORE the collective ...
    cb(buf.data_ptr(), buf.numel(), side.cuda_stream, None)      # ... the hook as the library calls it: raw pointer, count, stream handle
    with torch.cuda.stream(side):
        buf.add_(1.0)                                            # ... and AFTER it
    side.synchronize()
    np.save(out_path, buf.cpu().numpy())
    assert calls == [1000]
    st.close()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_the_allreduce_hook_runs_rccl_on_the_library_stream(tmp_path):
    """The hook the library calls five times per iteration, with the REAL backend (nccl = RCCL) in a one-rank group: a raw device pointer seen
    through the CUDA array interface, the collective issued on the stream handle the library passes, ordered between the work queued on that
    stream before and after it.  (World 1: the sum of one rank -- what is tested is that RCCL accepts the buffer and the external stream.)"""
    import torch.multiprocessing as mp
    out = str(tmp_path / "hook.npy")
    mp.spawn(_rccl_hook_rank, args=(1, _free_port(), out), nprocs=1, join=True)
    assert np.array_equal(np.load(out), 2.0 * np.arange(1000) + 1.0)
