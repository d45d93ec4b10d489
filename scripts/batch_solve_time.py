"""C4 banded solve: block cyclic reduction vs the sequential banded Cholesky (K = 2000, band 6 and 12)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from glio_amd import batch
for K, band, per_kf in ((2000, 6, 512), (2000, 12, 512)):
    gt, init = batch.make_poses(K)
    ci, cj, cp, nc, score = batch.make_constraints(gt, 0, K, per_kf, band, device="cuda:0")
    st = batch.BatchStage(K, band, len(ci)); st.set_constraints(ci, cj, cp, nc, score)
    Hg = st.new_hg(); st.linearize(init, Hg)
    st.set_solver(1); t1 = min(st.time_solve(Hg, 1e-4, 5) for _ in range(3)); n1, m1 = st.step(Hg, 1e-4, init)
    st.set_solver(0); t0 = st.time_solve(Hg, 1e-4, 2); n0, m0 = st.step(Hg, 1e-4, init)
    print(f"K {K} band {band}: block cyclic reduction {t1:.3f} ms, sequential {t0:.3f} ms, max |d step| {np.abs(n1 - n0).max():.2e}")
    st.close()
