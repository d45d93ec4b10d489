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


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    return b


def _no_constants(name):
    raise AssertionError("non-strict JSON constant in the bench line: " + name)


def test_the_stdout_line_is_small_strict_json():
    """round 5's driver record had `parsed: null`: the line had grown to 20 kB.  The stdout line is assembled by compact_line(): strict JSON
    (no NaN / Infinity), under 6 kB whatever the full record holds, with the contract's keys and the `roofline` / `cpu_baseline` objects."""
    b = _bench_module()
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_line.json")))
    full["pose_vs_oracle"]["max_rot_rad"] = float("nan")              # a NaN anywhere must not reach stdout
    full["roofline"]["others"]["junk"] = ["x" * 100] * 200             # nor may a section that grows
    full["released_config"] = {"cycle_ms": 0.5, "solve_ms": 0.1, "whatever": "y" * 5000}
    s = b.compact_line(full, "bench_full.json")
    assert len(s) < 6144 and "\n" not in s
    d = json.loads(s, parse_constant=_no_constants)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "iterations", "roofline", "cpu_baseline", "pose_vs_oracle", "whole_function_per_keyframe", "speedup_vs_cpu_port"):
        assert k in d, k
    assert d["pose_vs_oracle"]["max_rot_rad"] is None
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"], k
    assert "others" not in d["roofline"] and "large_launch" not in d["roofline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in d["cpu_baseline"], k
    assert "model" not in d["config"] and "workload" in d["config"]
    assert d["released_config"] == {"cycle_ms": 0.5, "solve_ms": 0.1}
    # a record so large that even the optional sections do not fit: they go, the contract's keys stay
    full["whole_function_per_keyframe"]["stages_ms"] = {f"stage_{i}": 0.123456 for i in range(250)}
    s2 = b.compact_line(full, "bench_full.json")
    assert len(s2) < 6144
    d2 = json.loads(s2, parse_constant=_no_constants)
    assert d2["roofline"]["frac"] and d2["cpu_baseline"]["value"] and "kernels_us" not in d2 and d2["whole_function_per_keyframe"] is None


def test_n2_line_carries_the_batch_stage_at_top_level_and_no_projection():
    """N > 1: the number a scaling record is about is the batch stage's strong-scaling time (the sliding window only replicates), so the compact line
    holds it at the top level together with the number of ranks RCCL saw; the 8-rank projection is not part of it."""
    line = _run(None, "--gpus", "2", "--dry-run")
    bs = line["batch_stage"]
    for k in ("scaling", "ranks", "rccl_ranks_seen", "ms_solve", "ms_solve_per_group", "kernel_groups", "constraints_total"):
        assert k in bs, k
    assert bs["ranks"] == 2 and bs["scaling"] == "strong" and bs["rccl_ranks_seen"] == 0
    assert "projection" not in json.dumps(line) and "projected" not in json.dumps(line)
