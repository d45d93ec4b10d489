"""First-contact diagnostics on the GPU box: per-stage parity numbers + rough timings."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from glio_amd import synth, capi
from oracle import pyoracle as po

def rel(a, b): return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)

print("devices", capi.device_count())
win = synth.make_window(W=4, pts_per_scan=600, with_gnss=True, with_prior=True)
corr = []
for s in range(win.W):
    q2, t2 = po.lidar_pose_for_association(win.opts, win.init.quat[s], win.init.trans[s])
    pts, pl, sc, _ = po.associate(win.opts, win.map_pts, win.scans[s], q2, t2)
    corr.append((pts, pl, sc))
for name, kw in [("lidar", dict(use_imu=False, use_gnss=False, use_prior=False)), ("+imu", dict(use_imu=True, use_gnss=False, use_prior=False)),
                 ("+prior", dict(use_imu=True, use_gnss=False, use_prior=True)), ("all", dict(use_imu=True, use_gnss=True, use_prior=True))]:
    prob = po.Problem(win, corr, **kw)
    ctx = capi.Context(win.opts)
    ctx.load_window(win, corr, **kw)
    st = win.init.copy()
    if not kw["use_gnss"]: st.n_ddt = 0
    Ho, go, co = prob.linearize(st)
    Hh, gh, ch = ctx.linearize(st)
    print(f"[{name}] cost {co:.6f} vs {ch:.6f} relH {rel(Hh,Ho):.2e} relg {rel(gh,go):.2e}")
    if rel(Hh, Ho) > 1e-9:
        D = np.abs(Hh - Ho); i, j = np.unravel_index(np.argmax(D), D.shape); print("  worst", i, j, Hh[i, j], Ho[i, j])
        blk = np.array([[np.abs(D[15*a:15*a+15, 15*b:15*b+15]).max() for b in range(win.W)] for a in range(win.W)]); print(blk)
    so, su_o = prob.solve(st)
    t = time.time(); sh, su_h = ctx.solve(st); dt = time.time() - t
    print(f"   solve oracle {su_o.as_dict()}\n   solve hip    {su_h.as_dict()}  wall {dt*1e3:.2f} ms")
    print("   dtrans", np.linalg.norm(sh.trans - so.trans, axis=1).max(), "dquat", np.abs(sh.quat - so.quat).max())
    ctx.close()

# bench shape
for (W, N, gn) in [(10, 16384, False), (20, 65536, True)]:
    t = time.time()
    win = synth.make_window(W=W, pts_per_scan=N, with_gnss=gn, with_prior=gn, seed=synth.SEED_BASE + 12)
    corr = synth.analytic_correspondences(win)
    print(f"W={W} N={N} gen {time.time()-t:.1f}s n_ddt={win.init.n_ddt} nres={sum(len(c[2]) for c in corr)}")
    ctx = capi.Context(win.opts)
    ctx.load_window(win, corr)
    prob = po.Problem(win, corr)
    t = time.time(); Ho, go, co = prob.linearize(win.init); t_or = time.time() - t
    Hh, gh, ch = ctx.linearize(win.init)
    print(f"  linearize: oracle {t_or*1e3:.1f} ms; cost {co:.4f}/{ch:.4f} relH {rel(Hh,Ho):.2e} relg {rel(gh,go):.2e}")
    for which, nm in [(0, "lidar_linearize"), (1, "full_linearize"), (2, "tr_step")]:
        ms = ctx.time_kernel(which, 20)
        extra = ""
        if which == 0:
            nres = sum(len(c[2]) for c in corr); extra = f"  -> {nres*40/ms/1e6:.1f} GB/s algorithmic"
        print(f"  {nm}: {ms*1e3:.1f} us{extra}")
    t = time.time(); so, su_o = prob.solve(win.init); t_os = time.time() - t
    sh, su_h = ctx.solve(win.init)
    t = time.time(); sh, su_h = ctx.solve(win.init); t_hs = time.time() - t
    ms, su = ctx.time_solve(win.init, 5)
    print(f"  solve: oracle {t_os*1e3:.1f} ms ({su_o.iterations} it) hip wall {t_hs*1e3:.2f} ms events {ms:.3f} ms ({su_h.iterations} it, term {su_h.termination})")
    print("  dtrans", np.linalg.norm(sh.trans - so.trans, axis=1).max(), "dquat", np.abs(sh.quat - so.quat).max())
    ctx.close()

# ---- isolated Cholesky check + association check
import ctypes as C
from glio_amd import ctypes_types as T
for n in (15, 16, 17, 150, 376, 700):
    o = synth.default_opts(W=50, pts=64, map_pts=64, n_ddt=0)
    ctx = capi.Context(o)
    rng = np.random.default_rng(n)
    B = rng.normal(size=(n, n)); A = B @ B.T + n * np.eye(n); b = rng.normal(size=n)
    x = np.zeros(n)
    rc = capi.load().glio_debug_chol_solve(ctx._h, n, T.dptr(np.ascontiguousarray(A)), T.dptr(b), T.dptr(x))
    print("chol n", n, "rc", rc, "err", np.abs(x - np.linalg.solve(A, b)).max() / np.abs(x).max())
    ctx.close()
win = synth.make_window(W=3, pts_per_scan=4000, seed=synth.SEED_BASE + 5)
ctx = capi.Context(win.opts)
t = time.time(); ctx.set_map(win.map_pts); print("set_map", len(win.map_pts), time.time() - t)
for s in range(win.W):
    q2, t2 = po.lidar_pose_for_association(win.opts, win.init.quat[s], win.init.trans[s])
    pts, pl, sc, src, nn = po.associate(win.opts, win.map_pts, win.scans[s], q2, t2, want_nn=True)
    t = time.time(); cnt = ctx.associate(s, win.scans[s], q2, t2); dt = time.time() - t
    hp, hpl, hsc = ctx.get_correspondences(s)
    print(f"assoc slot {s}: oracle {len(sc)} hip {cnt} ({dt*1e3:.2f} ms)", "pts eq", np.array_equal(hp, pts) if cnt == len(sc) else None,
          "planes maxdiff", np.abs(hpl - pl).max() if cnt == len(sc) else None, "scores maxdiff", np.abs(hsc - sc).max() if cnt == len(sc) else None)
    hnn = np.zeros((len(win.scans[s]), 5), np.int32)
    capi.load().glio_debug_last_nn(ctx._h, T.iptr(hnn), len(hnn))
    gate = hnn[:, 4] >= 0
    print("   nn equal where gate passes:", np.array_equal(hnn[gate], nn[gate]), "gate count", gate.sum())
print("map build ms", ctx.time_kernel(4, 10), "associate ms", ctx.time_kernel(3, 10))
ctx.close()
