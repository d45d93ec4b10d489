"""One steady-state window (W = 20, keyframe-chain step with helper workgroups) solved three times; prints a hash of every solve's result.  Run under
GLIO_CHAIN_HELPER_POLLS=0 (workgroup 0 gives its helpers up at once and sums the blocks itself), under HSA_CU_MASK (fewer CUs than the launch has
workgroups) and plainly, the hashes must agree: tests/test_hip_contention.py::test_chain_step_does_not_depend_on_its_helpers."""
import hashlib
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glio_amd import synth, capi  # noqa: E402
W, pts = 20, int(os.environ.get("HB_PTS", "4096"))
stream = synth.make_window(W=W + 1, pts_per_scan=pts, with_gnss=True, seed=synth.SEED_BASE + 12)
first = synth.sub_window(stream, 0, W)
c0 = capi.Context(first.opts); c0.load_window(first, synth.analytic_correspondences(first))
s0, _ = c0.solve(first.init); prior = c0.marginalize(s0); c0.close()
win = synth.sub_window(stream, 1, W); win.prior = prior
ctx = capi.Context(win.opts); ctx.load_window(win, synth.analytic_correspondences(win))
hashes, its = [], []
t0 = time.perf_counter()
for _ in range(3):
    sol, summ = ctx.solve(win.init)
    h = hashlib.sha256()
    for a in (sol.trans, sol.quat, sol.speed_bias, sol.rcv_ddt):
        h.update(a.tobytes())
    hashes.append(h.hexdigest()[:16]); its.append(int(summ.iterations))
import ctypes as C
st = (C.c_longlong * 320)()
capi.load().glio_debug_arrow_stamps(ctx._h, st)
fat_steps = int(st[300])
tr_us = round(ctx.time_kernel(capi.KERNEL_TR_STEP, 20) * 1e3, 2)
ms, _ = ctx.time_solve(win.init, 20)
print(json.dumps({"fat": os.environ.get("GLIO_CHAIN_FAT"), "fat_steps": fat_steps, "tr_step_us": tr_us, "solve_ms": round(ms, 4), "path": int(capi.load().glio_debug_solver_path(ctx._h)), "iterations": its, "hashes": hashes, "seconds": round(time.perf_counter() - t0, 3),
                  "polls": os.environ.get("GLIO_CHAIN_HELPER_POLLS"), "cu_mask": os.environ.get("HSA_CU_MASK")}))
