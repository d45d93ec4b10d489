"""Multi-window streaming parity: solve -> marginalize -> slide, repeated, HIP path vs the oracle running the
same call sequence (reference Estimator.cpp:2046-2736 once per keyframe).  Exercises association from the
evolving poses, the prior chain (device Cholesky root vs the oracle's eigen root) and the slot rotation."""
import numpy as np
import pytest

from glio_amd import sliding, synth

pytestmark = pytest.mark.gpu


class OracleBackend:
    """Same method set as capi.Context, computed by the CPU restatement (test infrastructure only)."""

    def __init__(self, opts):
        from oracle import pyoracle as po
        self.po, self.opts, self.W = po, opts, opts.window
        self.corr = [None] * self.W
        self.preints, self.prior, self.map = [], None, None

    def set_map(self, pts):
        self.map = pts

    def associate(self, slot, scan, q, t):
        pts, pl, sc, _ = self.po.associate(self.opts, self.map, scan, q, t)
        self.corr[slot] = (pts, pl, sc)
        return len(sc)

    def set_imu(self, preints):
        self.preints = preints

    def set_prior(self, prior):
        self.prior = prior

    def set_gnss(self, frame, dd, dop):
        assert frame is None

    def _problem(self):
        win = synth.Window(opts=self.opts, W=self.W, gt=None, init=None, kf_times=None, scans=None, scan_plane_id=None,
                           map_pts=self.map, scene=None, preints=self.preints, prior=self.prior)
        return self.po.Problem(win, self.corr, use_gnss=False, use_prior=self.prior is not None)

    def solve(self, state):
        return self._problem().solve(state)

    def marginalize(self, state):
        return self._problem().marginalize(state)


def rot_angle(qa, qb):
    d = synth.qmul(synth.qconj(qa), qb)
    return 2 * np.arctan2(np.linalg.norm(d[1:]), abs(d[0]))


def test_streaming_windows_match_oracle():
    from glio_amd import capi
    from oracle import pyoracle as po
    W, L = 4, 8                                    # 5 consecutive windows over 8 keyframes
    long = synth.make_window(W=L, pts_per_scan=700, seed=synth.SEED_BASE + 41)
    opts = synth.default_opts(W, pts=1024, map_pts=max(len(long.map_pts), 64))
    from glio_amd import ctypes_types as T
    first = T.WindowState(W)
    first.trans[:], first.quat[:], first.speed_bias[:] = long.init.trans[:W], long.init.quat[:W], long.init.speed_bias[:W]
    drivers = []
    ctx = capi.Context(opts)
    for be, lp in ((ctx, capi.lidar_pose), (OracleBackend(opts), po.lidar_pose_for_association)):
        d = sliding.SlidingWindowDriver(be, opts, lidar_pose=lp)
        d.start(first)
        drivers.append(d)
    for k in range(L - W + 1):
        outs = [d.step(long.map_pts, long.scans[k:k + W], long.preints[k:k + W - 1]) for d in drivers]
        (sh, smh, ch), (so, smo, co) = outs
        assert ch == co, f"window {k}: correspondence counts differ"
        assert smh.iterations == smo.iterations and smh.termination == smo.termination
        dt = np.linalg.norm(sh.trans - so.trans, axis=1).max()
        dr = max(rot_angle(sh.quat[i], so.quat[i]) for i in range(W))
        assert dt <= 1e-6 and dr <= 1e-7, f"window {k}: {dt:.2e} m {dr:.2e} rad"
        assert np.abs(sh.speed_bias - so.speed_bias).max() <= 1e-6
        # the solved window must sit near the ground truth (the stream is not drifting)
        assert np.linalg.norm(sh.trans - long.gt.trans[k:k + W], axis=1).max() < 0.2
        if k + W < L:
            for d in drivers:
                d.slide(long.init.trans[k + W], long.init.quat[k + W], long.init.speed_bias[k + W])
    assert drivers[0].first == L - W
    ctx.close()


def test_resident_stream_equals_roundtrip_stream():
    """Keeping scans and prior on the device between keyframes (slide_window / associate_window / marginalize_keep)
    reproduces the host-round-trip sequence bit for bit over five consecutive windows."""
    from glio_amd import capi
    from glio_amd import ctypes_types as T
    W, L = 4, 8
    long = synth.make_window(W=L, pts_per_scan=700, seed=synth.SEED_BASE + 41)
    opts = synth.default_opts(W, pts=1024, map_pts=max(len(long.map_pts), 64))
    first = T.WindowState(W)
    first.trans[:], first.quat[:], first.speed_bias[:] = long.init.trans[:W], long.init.quat[:W], long.init.speed_bias[:W]
    ca, cb = capi.Context(opts), capi.Context(opts)
    da = sliding.SlidingWindowDriver(ca, opts)
    db = sliding.ResidentSlidingWindow(cb, opts)
    da.start(first); db.start(first)
    for k in range(L - W + 1):
        sa, ma, na = da.step(long.map_pts, long.scans[k:k + W], long.preints[k:k + W - 1])
        sb, mb, nb = db.step(long.map_pts, long.scans[k:k + W], long.preints[k:k + W - 1])
        assert na == nb and ma.iterations == mb.iterations
        assert np.array_equal(sa.trans, sb.trans) and np.array_equal(sa.quat, sb.quat) and np.array_equal(sa.speed_bias, sb.speed_bias)
        if k + W < L:
            for d in (da, db):
                d.slide(long.init.trans[k + W], long.init.quat[k + W], long.init.speed_bias[k + W])
    ca.close(); cb.close()


def test_early_factor_uploads_equal_in_stream_uploads(monkeypatch):
    """glio_set_imu / glio_set_gnss while the window's searches run: by default the tables travel on a stream of their own into a device mirror and
    k_unstage (on the context's stream) installs them; GLIO_EARLY_UPLOAD=0 copies on the context's stream and waits.  Same solve, bit for bit --
    also when the calls come four in a row behind one search (another window's tables first: both upload blocks are sent again while the k_unstage
    that reads their mirrors is still queued behind the searches), and over three keyframes of a stream (marginalize, slide, set again)."""
    from glio_amd import capi
    long = synth.make_window(W=7, pts_per_scan=5000, seed=synth.SEED_BASE + 31, with_gnss=True)
    other = synth.make_window(W=4, pts_per_scan=64, seed=synth.SEED_BASE + 32, with_gnss=True)
    wins = [synth.sub_window(long, j, 4) for j in range(3)]
    out = []
    for early in ("0", "1"):
        monkeypatch.setenv("GLIO_EARLY_UPLOAD", early)
        o = wins[0].opts
        o.max_ddt_epochs = max(max(w.init.n_ddt for w in wins), other.init.n_ddt) + 4
        c = capi.Context(o)
        c.set_map(long.map_pts)
        c.set_prior(None)
        res = []
        for j, win in enumerate(wins):
            for s in range(win.W):
                c.set_scan(s, win.scans[s])
            poses = [capi.lidar_pose(win.opts, win.init.quat[s], win.init.trans[s]) for s in range(win.W)]
            c.associate_window_async(np.array([p[0] for p in poses]), np.array([p[1] for p in poses]))
            c.set_imu(other.preints); c.set_gnss(other.frame, other.dd, other.dop)
            c.set_imu(win.preints); c.set_gnss(win.frame, win.dd, win.dop)
            sol, summ = c.solve(win.init)
            res.append((summ.iterations, sol.trans.copy(), sol.quat.copy(), sol.speed_bias.copy(), summ.final_cost))
            c.marginalize_keep(sol)
        out.append(res)
        c.close()
    for a, b in zip(*out):
        assert a[0] == b[0] and a[4] == b[4]
        for x, y in zip(a[1:4], b[1:4]):
            assert np.array_equal(x, y)


@pytest.mark.gpu
def test_scan_sent_ahead_and_asynchronous_marginalization_change_nothing():
    """glio_set_scan_ahead (the next keyframe's scan into the ring row the next slide exposes, on the upload stream) and glio_marginalize_keep_async / _finish against
    the plain calls: two contexts walk the same three keyframes -- same counts, same correspondences, same solved states, same priors in effect."""
    import numpy as np
    from glio_amd import capi, synth
    W, NK = 4, 3
    long = synth.make_window(W=W + NK, pts_per_scan=3000, with_gnss=False, with_prior=False, seed=synth.SEED_BASE + 33)
    wins = [synth.sub_window(long, j, W) for j in range(NK + 1)]
    a, b = capi.Context(wins[0].opts), capi.Context(wins[0].opts)
    for c in (a, b):
        c.set_map(wins[0].map_pts)
        for s in range(W):
            c.set_scan(s, wins[0].scans[s])
        c.set_prior(None)
    sent = False
    for j in range(NK + 1):
        win = wins[j]
        poses = [capi.lidar_pose(win.opts, win.init.quat[s], win.init.trans[s]) for s in range(W)]
        q2s = np.array([p[0] for p in poses]); t2s = np.array([p[1] for p in poses])
        if j > 0:
            a.slide_window(); a.set_scan(W - 1, win.scans[W - 1])
            b.slide_window()
            if not sent:
                b.set_scan(W - 1, win.scans[W - 1])
        ca = a.associate_window(q2s, t2s)
        b.associate_window_async(q2s, t2s); cb = b.associate_window_counts()
        assert np.array_equal(ca, cb) and ca.sum() > 0, j
        for s in range(W):
            assert all(np.array_equal(x, y) for x, y in zip(a.get_correspondences(s), b.get_correspondences(s))), (j, s)
        for c in (a, b):
            c.set_imu(win.preints)
        sa, ma = a.solve(win.init); sb, mb = b.solve(win.init)
        assert ma.iterations == mb.iterations and np.array_equal(sa.trans, sb.trans) and np.array_equal(sa.quat, sb.quat), j
        a.marginalize_keep(sa)
        b.marginalize_keep_async(sb)
        sent = j < NK
        if sent:
            b.set_scan_ahead(wins[j + 1].scans[W - 1])
        if j % 2:
            b.marginalize_keep_finish()          # (else: the next solve finishes it by itself)
    a.close(); b.close()
