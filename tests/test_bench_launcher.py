"""`python bench.py --gpus N` must produce an N-rank job by itself when no launcher set WORLD_SIZE (the driver's contract also
allows `python -m torch.distributed.run ... bench.py --gpus N`, where the ranks come from the environment).  CPU: dry mode (the
ranks rendezvous over gloo and rank 0 prints the line; no GPU work)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env=None, *args):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(extra_env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_gpus_2_spawns_two_ranks():
    line = _run(None, "--gpus", "2", "--dry-run")
    assert line["n_gpus"] == 2 and line["rank_sum"] == 3.0


def test_gpus_3_spawns_three_ranks():
    line = _run(None, "--gpus", "3", "--dry-run")
    assert line["n_gpus"] == 3 and line["rank_sum"] == 6.0


def test_one_rank_needs_no_rendezvous():
    line = _run(None, "--dry-run")
    assert line["n_gpus"] == 1


def test_under_torch_distributed_run_the_ranks_come_from_the_environment():
    """the driver's launch line for N > 1: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...` -- bench.py must not spawn again, and exactly one JSON line must come out"""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["rank_sum"] == 3.0
