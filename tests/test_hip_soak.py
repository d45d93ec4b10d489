"""Soak and concurrency tests of the device-resident solver (the class of defect DESIGN.md calls the "coherence trap":
results that depend on what ran before).

* 300 back-to-back solves of one window on a reused context, and solves on a stream of fresh contexts, must be bit-identical
  (state, iteration count, costs) -- also with every CU's LDS NaN-poisoned before each kernel group and every context
  allocation filled with NaN bytes (GLIO_DEBUG_LDS_POISON / GLIO_DEBUG_FILL are read when the library loads, so that part
  runs in a child process).
* The reference drives the sliding-window problem and the batch problem from two threads at once
  (GLIO/src/Estimator.cpp:5398-5404, no lock at :2751): a sliding-window context and a batch context hammered from two
  threads, with a third creating and destroying contexts, must each produce what they produce alone."""
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

from glio_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _window(W=8):
    stream = synth.make_window(W=W + 1, pts_per_scan=3000 if W <= 8 else 800, with_gnss=True, with_prior=False, seed=synth.SEED_BASE + 41)
    return stream


def _digest(sol, summ):
    return (sol.trans.tobytes(), sol.quat.tobytes(), sol.speed_bias.tobytes(), sol.rcv_ddt.tobytes(), int(summ.iterations),
            int(summ.termination), float(summ.final_cost), float(summ.initial_cost))


_SOAK_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from glio_amd import capi, synth
from test_hip_soak import _window, _digest
W = %d
stream = _window(W)
first = synth.sub_window(stream, 0, W)
ctx0 = capi.Context(first.opts); ctx0.load_window(first, synth.analytic_correspondences(first))
sol0, _ = ctx0.solve(first.init)
prior = ctx0.marginalize(sol0); ctx0.close()
win = synth.sub_window(stream, 1, W); win.prior = prior            # steady state: the keyframe-chain solver path
corr = synth.analytic_correspondences(win)
ref = None
ctx = capi.Context(win.opts); ctx.load_window(win, corr)
for k in range(%d):
    d = _digest(*ctx.solve(win.init))
    ref = ref or d
    assert d == ref, f"reused context: solve {k} differs"
path = capi.load().glio_debug_solver_path(ctx._h)
fronts = capi.load().glio_debug_chain_fronts_used(ctx._h)
ctx.close()
for k in range(%d):
    c = capi.Context(win.opts); c.load_window(win, corr)
    for j in range(3):
        assert _digest(*c.solve(win.init)) == ref, f"fresh context {k}, solve {j} differs"
    c.close()
print("SOAK_OK", path, fronts, ref[4])
"""


@pytest.mark.parametrize("W,poison", [(8, False), (8, True), (20, False), (20, True), (28, False), (28, True)],
                         ids=["plain", "lds_poison_nan_fill", "four_fronts", "four_fronts_lds_poison_nan_fill", "global_blocks", "global_blocks_lds_poison_nan_fill"])
def test_repeated_solves_are_bit_identical(W, poison):
    """W = 20: k_chain_step with the separator and four fronts -- its hand-overs are flags in LDS polled by eight wavefronts, so the interleaving differs
    from run to run while every number must not; W = 28: the chain with its blocks in global memory (k_chain_solve<true>, band-only k_assemble)."""
    env = dict(os.environ)
    if poison:
        env["GLIO_DEBUG_LDS_POISON"] = "1"
        env["GLIO_DEBUG_FILL"] = "255"
    n_reuse, n_fresh = (300, 25) if not poison else (120, 10)
    if W > 8:
        n_reuse, n_fresh = n_reuse // 2, n_fresh // 2
    out = subprocess.run([sys.executable, "-c", _SOAK_SCRIPT % (ROOT, os.path.join(ROOT, "tests"), W, n_reuse, n_fresh)], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "SOAK_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
    f = out.stdout.split("SOAK_OK")[1].split()
    assert f[0] == "2", "the steady-state window must take the keyframe-chain path"
    assert f[1] == ("2" if W == 8 else "4"), "elimination fronts"


def test_sliding_window_and_batch_contexts_from_two_threads():
    import torch  # noqa: F401  (the HIP runtime of this process)
    from glio_amd import batch, capi
    stream = _window()
    win = synth.sub_window(stream, 0, 8)
    corr = synth.analytic_correspondences(win)
    ctx = capi.Context(win.opts); ctx.load_window(win, corr)
    ref_sw = _digest(*ctx.solve(win.init))

    K, band = 96, 6
    gt, init = batch.make_poses(K)
    ci, cj, cp, nc, score = batch.make_constraints(gt, 0, K, 512, band, device="cuda:0")
    st = batch.BatchStage(K, band, len(ci)); st.set_constraints(ci, cj, cp, nc, score)
    Hg = st.new_hg()
    st.linearize(init, Hg)
    ref_hg = Hg.cpu().numpy().copy()
    ref_step, ref_md = st.step(Hg, 1e-4, init)

    errors = []
    stop = threading.Event()

    def sw_thread():
        try:
            for _ in range(150):
                if _digest(*ctx.solve(win.init)) != ref_sw:
                    errors.append("sliding-window solve changed under concurrency"); return
        except Exception as e:  # noqa: BLE001
            errors.append(f"sw: {e}")

    def batch_thread():
        try:
            hg = st.new_hg()
            for _ in range(150):
                st.linearize(init, hg)
                if not np.array_equal(hg.cpu().numpy(), ref_hg):
                    errors.append("batch linearisation changed under concurrency"); return
                out, md = st.step(hg, 1e-4, init)
                if not np.array_equal(out, ref_step) or md != ref_md:
                    errors.append("batch step changed under concurrency"); return
        except Exception as e:  # noqa: BLE001
            errors.append(f"batch: {e}")

    def churn_thread():
        try:
            small = synth.sub_window(stream, 2, 4)
            sc = synth.analytic_correspondences(small)
            ref = None
            while not stop.is_set():
                c = capi.Context(small.opts); c.load_window(small, sc)
                d = _digest(*c.solve(small.init))
                ref = ref or d
                if d != ref:
                    errors.append("churned context differs"); return
                c.close()
        except Exception as e:  # noqa: BLE001
            errors.append(f"churn: {e}")

    ts = [threading.Thread(target=f) for f in (sw_thread, batch_thread)]
    tc = threading.Thread(target=churn_thread)
    tc.start()
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=600)
    stop.set(); tc.join(timeout=120)
    assert not errors, errors
    ctx.close(); st.close()


def test_one_context_through_changing_factor_structures_equals_fresh_contexts():
    """A context that is REUSED while the structure of its problem changes -- GNSS on / off, prior on / off (block structure changes with it), IMU
    on / off, a slot without correspondences, another window of the stream -- must return, every time, exactly what a fresh context returns for
    the same inputs: nothing of an earlier configuration (chain-layout slices, epoch tables, prior tables, look-ahead groups still draining) may
    leak into a later one.  Each configuration is solved twice on the reused context (the second solve finds the first one's result in every
    buffer; the first one does not) and marginalized."""
    from glio_amd import capi
    stream = _window()
    wins = [synth.sub_window(stream, lo, 8) for lo in (0, 1)]
    corr = [synth.analytic_correspondences(w) for w in wins]
    empty = lambda c: (c[0][:0], c[1][:0], c[2][:0])
    # a prior with real structure: the marginalization of window 0
    c0 = capi.Context(wins[0].opts); c0.load_window(wins[0], corr[0])
    s0, _ = c0.solve(wins[0].init); prior = c0.marginalize(s0); c0.close()
    wins[1].prior = prior
    configs = [
        ("window 1: everything", 1, dict(), None),
        ("window 0: no prior", 0, dict(), None),
        ("window 1: no GNSS", 1, dict(use_gnss=False), None),
        ("window 1: no prior, no IMU", 1, dict(use_prior=False, use_imu=False), None),
        ("window 1: slot 3 without correspondences", 1, dict(), 3),
        ("window 0: no GNSS, no prior", 0, dict(use_gnss=False), None),
        ("window 1: everything again", 1, dict(), None),
    ]
    reused = capi.Context(wins[0].opts)
    for name, wi, kw, drop in configs:
        w, cr = wins[wi], list(corr[wi])
        if drop is not None:
            cr[drop] = empty(cr[drop])
        fresh = capi.Context(w.opts); fresh.load_window(w, cr, **kw)
        sol_f, sm_f = fresh.solve(w.init)
        want = _digest(sol_f, sm_f)
        marg_f = fresh.marginalize(sol_f)
        fresh.close()
        reused.load_window(w, cr, **kw)
        for attempt in (1, 2):
            sol_r, sm_r = reused.solve(w.init)
            assert _digest(sol_r, sm_r) == want, f"{name}: solve {attempt} on the reused context differs from a fresh context"
        marg_r = reused.marginalize(sol_r)
        assert np.array_equal(marg_r["lin_jac"], marg_f["lin_jac"]) and np.array_equal(marg_r["lin_res"], marg_f["lin_res"]), f"{name}: marginalization differs"
    reused.close()
