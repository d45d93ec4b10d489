import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glio_amd import synth, capi
o = synth.default_opts(W=50, pts=64, map_pts=64, n_ddt=0)
ctx = capi.Context(o)
for n in (150, 376):
    for skip, nm in [(0, "full"), (8, "no backsub"), (9, "no mfma/backsub"), (10, "no diag/backsub"), (12, "no trsm/backsub"), (15, "barriers+loads only")]:
        ms = C.c_float()
        rc = capi.load().glio_debug_chol_time(ctx._h, n, 10, skip, C.byref(ms))
        print(f"n={n} {nm:22s}: {ms.value*1e3:7.1f} us (rc {rc})")
