"""glio_set_scan for one 65536-point scan (1 MB): the call's wall time from a pageable and from a pinned source buffer, GPU idle."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from glio_amd import synth, capi
o = synth.default_opts(4, pts=65536, map_pts=64)
ctx = capi.Context(o)
rng = np.random.default_rng(1)
scan = np.ascontiguousarray(rng.normal(0, 20, (65536, 4)).astype(np.float32))
pin = torch.empty((65536, 4), dtype=torch.float32).pin_memory()
pin.numpy()[:] = scan
def run(a, n=60):
    for _ in range(5): ctx.set_scan(1, a)
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); ctx.set_scan(1, a); ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e6
    return round(float(np.median(ts)), 1), round(float(ts.min()), 1)
print("pageable source: median, min us", run(scan))
print("pinned source:   median, min us", run(pin.numpy()))
for npts in (4096, 16384, 32768):
    print(npts, "points pageable", run(scan[:npts]), "pinned", run(pin.numpy()[:npts]))
