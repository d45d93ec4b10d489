import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glio_amd import synth, capi
win = synth.make_window(W=20, pts_per_scan=65536, with_gnss=True, with_prior=True, seed=synth.SEED_BASE + 12)
corr = synth.analytic_correspondences(win)
ctx = capi.Context(win.opts)
ctx.load_window(win, corr)
for _ in range(3):
    ctx.solve(win.init)
    time.sleep(0.01)
