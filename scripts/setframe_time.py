import sys, time
sys.path.insert(0, ".")
import numpy as np
from glio_amd import batch, synth
win = synth.make_window(W=4, pts_per_scan=32768, seed=synth.SEED_BASE + 61, perturb=(0.03, 0.2, 0.0), scan_radius=25.0, map_density=0.5)
K = 200
ba = batch.BatchAssociation(K, 32768, 1000000)
for k in range(8): ba.set_frame(k, win.scans[k % 4])
t0 = time.perf_counter()
for k in range(K): ba.set_frame(k, win.scans[k % 4])
print("set_frame ms per frame", (time.perf_counter() - t0) / K * 1e3)
ba.close()
