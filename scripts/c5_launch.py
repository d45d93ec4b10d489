"""Workload of scripts/c5_pmc.sh: the C5-shape launch of K3 in its fp32-Jacobian / MFMA form (50 keyframes x 262 144
residuals, 32 B each = 419 MB per launch) and, for comparison, the fp64 form on the same records (524 MB)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from glio_amd import capi, synth
from glio_amd import ctypes_types as T

W, P = 50, 262144
rng = np.random.default_rng(5)
p = np.zeros((P, 4), np.float32); p[:, :3] = rng.uniform(-30, 30, (P, 3))
n = rng.normal(0, 1, (P, 3)); n /= np.linalg.norm(n, axis=1, keepdims=True)
pl = np.zeros((P, 4), np.float32); pl[:, :3] = 0.8 * n; pl[:, 3] = rng.uniform(-5, 5, P)
sc = rng.uniform(3, 7.5, P)
for prec in (1, 0):
    o = synth.default_opts(W, pts=P, map_pts=64)
    o.lidar_precision = prec
    ctx = capi.Context(o)
    for s in range(W):
        ctx.set_correspondences(s, np.roll(p, s, axis=0), pl, sc)
    ctx.set_imu([]); ctx.set_prior(None); ctx.set_gnss(None, [], [])
    st = T.WindowState(W)
    ctx.linearize(st, want_H=False)
    reps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 10
    if "--sweep" in sys.argv:          # launch geometry sweep (workgroups per keyframe), informational
        for bpk in (8, 15, 30, 60, 120, 240):
            capi.load().glio_debug_set_k3(ctx._h, bpk, 24 if prec == 0 else 22)
            ms = min(ctx.time_kernel(capi.KERNEL_LIDAR_LINEARIZE, reps) for _ in range(2))
            rd = min(ctx.time_kernel(capi.KERNEL_STREAM_READ, reps) for _ in range(2))
            print(f"precision {prec} bpk {bpk:4d}: {ms * 1e3:7.1f} us per launch, {W * P * (32 if prec else 40) / ms / 1e6:6.0f} GB/s; read-only {W * P * (32 if prec else 40) / rd / 1e6:6.0f} GB/s")
    else:
        if prec == 0:
            capi.load().glio_debug_set_k3(ctx._h, 15, 24)
        ms = ctx.time_kernel(capi.KERNEL_LIDAR_LINEARIZE, reps)
        print(f"precision {prec}: {ms * 1e3:.1f} us per launch, {W * P * (32 if prec else 40) / ms / 1e6:.0f} GB/s")
    ctx.close()
