"""Several PROCESSES sharing the GPU, each creating fresh contexts in a loop: solve twice, marginalize.  The first solve of a context
reads its result from host-mapped memory that has never held a result before -- the case in which round 3 found the completion tag
visible ahead of parts of the payload (whole keyframes of the returned state still zero, `done` still 0; ~10 % of the fresh contexts with
12 processes on one MI355X, none with a single process).  The payload now carries a checksum that the host verifies (glio_device.h,
SolverStatus::checksum); every iteration here must give the same result, and the first solve must equal the second."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
def test_fresh_contexts_under_contention_return_complete_results():
    env = dict(os.environ, REPRO_TWICE="1", REPRO_PTS="16384")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "contention_loop.py"), "8", "10"], env=env, capture_output=True, text=True, timeout=540)
    rows = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(rows) == 8, out.stdout[-2000:] + out.stderr[-2000:]
    for r in rows:
        assert r["events"] == [] and r["distinct"] == 1, r


@pytest.mark.timeout(600)
def test_association_local_map_and_batch_solve_are_bit_stable_under_contention():
    """The same load on the other entry points, fresh objects every iteration: K1 + K2 association records, the local map, the batch
    problem's trust-region solve (IMU chain) -- hashes of the outputs must not change from iteration to iteration.  (Round 3: about 1 % of the
    first associations of a fresh context kept a third fewer correspondences -- the query-binning table was initialised by a NULL-stream
    hipMemset, which is not ordered with the library's non-blocking stream and, under load, ran after the table had been filled.)"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "contention_more.py"), "8", "12"], capture_output=True, text=True, timeout=540)
    rows = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(rows) == 8, out.stdout[-2000:] + out.stderr[-2000:]
    for r in rows:
        assert r["events"] == [] and r["distinct"] == 1, r


@pytest.mark.timeout(600)
def test_the_resident_keyframe_stream_is_deterministic_under_contention():
    """The whole per-keyframe sequence on resident data (slide, new scan, local-map push of the resident scan, asynchronous association with the factor
    tables staged meanwhile, solve, marginalize-and-keep) over six keyframes, run three times per process on fresh contexts with six processes on the
    GPU: every run must reproduce the first bit for bit."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "contention_stream.py"), "6", "2"], capture_output=True, text=True, timeout=540)
    rows = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(rows) == 6, out.stdout[-2000:] + out.stderr[-2000:]
    for r in rows:
        assert r["events"] == [], r


@pytest.mark.timeout(600)
def test_chain_step_does_not_depend_on_its_helpers():
    """k_chain_step's workgroup 0 waits INSIDE the launch for W helper workgroups -- a wait HIP does not promise to be satisfiable.  It is bounded: after
    `hpolls` polls the step sums the blocks itself.  Same bits out (a) plainly, (b) with the helpers given up at once (GLIO_CHAIN_HELPER_POLLS=0), (c) on
    a device that offers the launch only 2 compute units (HSA_CU_MASK: 21 workgroups, each a CU's worth of LDS), (d) both -- and none of them hangs."""
    def run(extra):
        env = dict(os.environ, **extra)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "chain_helpers_bound.py")], env=env, capture_output=True, text=True, timeout=240)
        rows = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
        assert len(rows) == 1, out.stdout[-2000:] + out.stderr[-2000:]
        return rows[0]
    ref = run({})
    assert ref["path"] == 2 and len(set(ref["hashes"])) == 1, ref
    # (round 6: the helpers are "fat" -- they build every keyframe's block, rows of t and epoch columns speculatively; every step of these solves takes
    #  their products, 4 per solve; with GLIO_CHAIN_FAT=0 they only sum, with the waits given up workgroup 0 builds everything itself: the same bits)
    assert ref["fat_steps"] == sum(ref["iterations"]), ref
    for extra in ({"GLIO_CHAIN_FAT": "0"}, {"GLIO_CHAIN_HELPERS": "0"}, {"GLIO_CHAIN_HELPER_POLLS": "0"}, {"HSA_CU_MASK": "0:0-1"},
                  {"HSA_CU_MASK": "0:0-1", "GLIO_CHAIN_HELPER_POLLS": "3"}):
        got = run(extra)
        assert got["path"] == 2 and got["hashes"] == ref["hashes"] and got["iterations"] == ref["iterations"], (extra, got, ref)
        if "GLIO_CHAIN_FAT" in extra or "GLIO_CHAIN_HELPERS" in extra or extra.get("GLIO_CHAIN_HELPER_POLLS") == "0":
            assert got["fat_steps"] == 0, (extra, got)


@pytest.mark.timeout(600)
def test_a_10_hz_caller_sees_no_stall(tmp_path):
    """Round 5 left an unexplained 11-25 ms marginalization stage in the Python stream driver when the batch association was PREPARED before the solve and
    seconds of host work followed (NOTES_r05.md).  The C++ host uses exactly that order, and a real caller runs at ~10 Hz: ~100 ms of nothing between calls.
    host_demo_stream with the host sleeping 100 ms inside every keyframe call -- between the preparation and the solve, and between the solve and the
    association's enqueue: no stage of any keyframe may take more than twice its warm time (+ 0.25 ms for the clocks an idle GPU lets drop).
    (Round 6: not reproduced in either host with sleeps of 20 ms - 2 s, busy loops, 2 GB of allocation churn or a CPU oracle solve in the gap;
    scripts/stall_repro.py, scripts/stall_py.py.  What an idle gap does cost is the clock ramp: 0.30 -> 0.47 ms for the marginalization after 2 s.)"""
    from glio_amd import synth
    from glio_amd.host import window_io
    W, pts, NK = 20, 16384, 5
    long = synth.make_window(W=W + NK, pts_per_scan=pts, with_gnss=True, with_prior=False, seed=synth.SEED_BASE + 12)
    wins = [synth.sub_window(long, j, W) for j in range(NK + 1)]
    opts = wins[0].opts
    opts.max_ddt_epochs = max(w.init.n_ddt for w in wins) + 8
    opts.max_map_points = 1 << 18
    path = str(tmp_path / "s.bin")
    window_io.write_stream(path, long, wins, W, NK, pts)
    window_io.run_demo_stream(path)
    warm = window_io.run_demo_stream(path)
    wavg = list(warm["stages_ms"].values())
    for at in (0, 2):
        got = window_io.run_demo_stream(path, sleep_ms=100, sleep_at=at)
        assert got["iterations"] == warm["iterations"] and got["trans_checksum"] == warm["trans_checksum"]
        for name, mx, w in zip(warm["stages_ms"], got["stage_max_ms"], wavg):
            assert mx <= 2.0 * w + 0.25, (at, name, mx, w, got["stage_max_ms"])
