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


def test_k2_window_block_reads_the_newest_committed_profiles():
    """bench.py's K2 roofline block takes its kernel times and VALU counts from profiles/rNN_k2_window_* (advisor, round 4: no constants of an old
    profile in the bench line): the helper finds the newest pair and derives the VALU-issue floor from it."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    w = b.k2_window_block({"window_associate_one_call_ms": 0.32, "queries_per_scan": 65536}, 20)
    assert w is not None and w["kernel_time_source"].endswith("_k2_window_kernel_stats.csv")
    k = w["kernels_us_per_window_call"]
    assert k["k_knn5_near<64>"] > 10 and k["k_plane_fit<false>"] > 5 and 0.0 < w["frac"] < 1.0
    v = w["valu_issue"]
    assert v["wave_valu_instructions_per_window_call"]["k_knn5_near<64>"] > 1e6 and 0.0 < v["frac_of_valu_issue_search_kernels"] <= 1.0
