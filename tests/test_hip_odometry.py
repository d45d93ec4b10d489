"""GPU parity of the front-end scan-to-map odometry against the oracle running the same sequence
(reference GLIO/src/LidarOdometry.cpp:343-404, 474-581): association with the front end's gates, unit-score plane
factors under Huber(0.1), Levenberg-Marquardt with Ceres-1.14 radius control."""
import numpy as np
import pytest

from glio_amd import odometry, synth
from glio_amd import ctypes_types as T

pytestmark = pytest.mark.gpu


class OracleBackend:
    def __init__(self, opts):
        from oracle import pyoracle as po
        self.po, self.opts = po, opts
        self.map = self.scan = self.corr = None

    def set_map(self, m): self.map = m
    def set_scan(self, slot, scan): self.scan = scan
    def set_imu(self, p): pass
    def set_prior(self, p): pass
    def set_gnss(self, f, a, b): pass

    def associate_resident(self, slot, q, t):
        pts, pl, sc, _ = self.po.associate(self.opts, self.map, self.scan, q, t)
        self.corr = [(pts, pl, sc)]
        return len(sc)

    def solve(self, state):
        win = synth.Window(opts=self.opts, W=1, gt=None, init=None, kf_times=None, scans=None, scan_plane_id=None, map_pts=self.map, scene=None)
        return self.po.Problem(win, self.corr, use_gnss=False, use_prior=False, use_imu=False).solve(state)


@pytest.fixture(scope="module")
def frame():
    win = synth.make_window(W=1, pts_per_scan=3000, seed=synth.SEED_BASE + 71, perturb=(0.15, 0.8, 0.0), scan_radius=30.0)
    scan = win.scans[0].copy()
    scan[:, :3] -= np.array(win.opts.t_lb, np.float32)          # body-frame cloud: the front end has no extrinsic
    pose0 = np.r_[win.init.quat[0], win.init.trans[0]]
    gt = np.r_[win.gt.quat[0], win.gt.trans[0]]
    return win.map_pts, np.ascontiguousarray(scan), pose0, gt


def test_unit_scores_and_gates(frame):
    from glio_amd import capi
    mp, scan, pose0, _ = frame
    o = odometry.frontend_opts(len(scan), len(mp))
    ctx = capi.Context(o)
    ctx.set_map(mp); ctx.set_scan(0, scan)
    n = ctx.associate_resident(0, pose0[:4], pose0[4:])
    pts, planes, scores = ctx.get_correspondences(0)
    assert n == len(scores) > 300
    assert np.all(scores == 1.0)                                     # LidarPlaneNormIncreFactor carries no score
    w = np.linalg.norm(planes[:, :3].astype(np.float64), axis=1)     # |w n| = w
    assert w.min() > 0.4 - 1e-6 and w.max() <= 1.0 + 1e-6            # LidarOdometry.cpp:392
    ob = OracleBackend(o); ob.set_map(mp); ob.set_scan(0, scan)
    assert ob.associate_resident(0, pose0[:4], pose0[4:]) == n
    assert np.array_equal(ob.corr[0][1], planes) and np.array_equal(ob.corr[0][0], pts)
    ctx.close()


@pytest.mark.parametrize("match_cnt", [1, 3])
def test_scan_to_map_matches_oracle(frame, match_cnt):
    from glio_amd import capi
    mp, scan, pose0, gt = frame
    o = odometry.frontend_opts(len(scan), len(mp))
    ctx = capi.Context(o)
    res = []
    for be in (ctx, OracleBackend(o)):
        od = odometry.ScanToMapOdometry(be)
        od.set_map(mp)
        res.append(od.update(scan, pose0, match_cnt=match_cnt))
    (ph, rh), (po_, ro) = res
    for (sh, kh), (so, ko) in zip(rh, ro):
        assert kh == ko
        assert sh.iterations == so.iterations and sh.termination == so.termination and sh.successful_steps == so.successful_steps
        assert abs(sh.final_cost - so.final_cost) <= 1e-9 * abs(so.final_cost)
    assert np.abs(ph - po_).max() <= 1e-9
    # the estimate moves towards the ground truth
    assert np.linalg.norm(ph[4:] - gt[4:]) < 0.3 * np.linalg.norm(pose0[4:] - gt[4:])
    ctx.close()


def test_solver_time_budget_ends_the_solve_at_an_accepted_point(frame):
    """options.max_solver_time_in_seconds (LidarOdometry.cpp:524: 0.015 s).  With the front end's budget the solve is untouched (it
    takes ~0.3 ms); with a budget that is spent at once the device loop stops at the next iteration start with NO_CONVERGENCE, at
    a point it had accepted (cost not above the initial cost), and later solves on the same context are not affected."""
    from glio_amd import capi
    map_pts, scan, pose0, gt = frame
    outs = {}
    for budget in (0.015, 1e-7):
        o = odometry.frontend_opts(len(scan), len(map_pts))
        o.max_solver_time_s = budget
        ctx = capi.Context(o)
        odo = odometry.ScanToMapOdometry(ctx)
        odo.set_map(map_pts)
        pose, rounds = odo.update(scan, pose0, match_cnt=1)
        outs[budget] = (pose, rounds[0][0])
        if budget < 1e-3:      # the same context again with the normal budget: unaffected by the stop word of the earlier solve
            ctx2 = capi.Context(odometry.frontend_opts(len(scan), len(map_pts)))
            odo2 = odometry.ScanToMapOdometry(ctx2); odo2.set_map(map_pts)
            pose2, rounds2 = odo2.update(scan, pose0, match_cnt=1)
            assert rounds2[0][0].iterations == outs[0.015][1].iterations and np.abs(pose2 - outs[0.015][0]).max() == 0.0
            ctx2.close()
        ctx.close()
    full, cut = outs[0.015][1], outs[1e-7][1]
    assert full.termination != 0 and full.iterations >= 3
    assert cut.termination == 0 and cut.iterations < full.iterations                  # GLIO_TERM_NO_CONVERGENCE
    assert cut.final_cost <= cut.initial_cost * (1 + 1e-12) and np.all(np.isfinite(outs[1e-7][0]))


def _scan_stream(n_scans=8, pts=3000, seed=synth.SEED_BASE + 73):
    """consecutive LiDAR frames of one drive (body frame, as surf_last_ds), expressed relative to the FIRST frame's pose: the front end starts at identity"""
    win = synth.make_window(W=n_scans, pts_per_scan=pts, seed=seed, scan_radius=30.0, kf_dt=0.1)
    tlb = np.array(win.opts.t_lb, np.float32)
    scans = []
    for s in range(n_scans):
        c = win.scans[s].copy(); c[:, :3] -= tlb
        scans.append(np.ascontiguousarray(c))
    return win, scans


class OracleRingBackend(OracleBackend):
    """OracleBackend + the 20-frame local map: clouds at their poses (orc_transform_cloud), concatenated in ring order, pcl::VoxelGrid (orc_voxel_grid)"""

    def localmap_config(self, width, leaf, cap):
        self.width, self.leaf, self.ring = width, leaf, []

    def localmap_push(self, cloud, q, t):
        self.ring.append(self.po.transform_cloud(cloud, q, t))
        self.ring = self.ring[-self.width:]

    def localmap_build(self):
        self.map, _ = self.po.voxel_grid(np.vstack(self.ring), self.leaf)
        return len(self.map)


def test_front_end_run_follows_the_oracle_over_a_stream():
    """LidarOdometry::run() per scan (LidarOdometry.cpp:661-699: poseInitialization, buildLocalMap with the 20-frame ring, downSampleCloud at 0.2 m,
    updateTransformationWithCeres, savePoses, computeRelative) on the device-resident local map (pcl's float accumulation: bit-identical maps) against the same
    loop on the oracle: the same rounds, kept counts and iterations scan after scan, poses to 1e-9 although every scan starts from the previous result."""
    from glio_amd import capi
    win, scans = _scan_stream()
    o = odometry.frontend_opts(max(len(s) for s in scans), 1 << 16)
    ctx = capi.Context(o)
    od_h, od_o = odometry.ScanToMapOdometry(ctx), odometry.ScanToMapOdometry(OracleRingBackend(o))
    for k, sc in enumerate(scans):
        if k == 1:
            ctx.localmap_set_accumulation(1)              # (the ring exists from the first run() on)
        ph, rh = od_h.run(sc, max_points=o.max_points_per_scan)
        po_, ro = od_o.run(sc, max_points=o.max_points_per_scan)
        assert len(rh) == len(ro) == (0 if k == 0 else (8 if k == 1 else 1)), k
        for (sh, kh), (so, ko) in zip(rh, ro):
            assert kh == ko and sh.iterations == so.iterations and sh.termination == so.termination, k
        assert np.abs(ph - po_).max() <= 1e-9, (k, np.abs(ph - po_).max())
        assert od_h.map_points == od_o.map_points
        assert np.abs(od_h.rel_pose - od_o.rel_pose).max() <= 1e-9
    # the drive moves ~1 m per scan (synthetic 10 m/s at 10 Hz); the estimate follows it: relative motion of the last step vs the ground truth's
    gt_rel = np.linalg.norm(win.gt.trans[-1] - win.gt.trans[-2])
    assert abs(np.linalg.norm(od_h.rel_pose[4:]) - gt_rel) < 0.1 * gt_rel + 0.05
    ctx.close()


def test_cpp_front_end_equals_the_python_twin(tmp_path):
    """glio::ScanToMapOdometry (glio_backend.hpp, driven by host_demo_odometry) against glio_amd/odometry.py on the same scans through the same C entry
    points: bit-identical poses, the same rounds / kept / iterations / map sizes per scan."""
    from glio_amd import capi
    from glio_amd.host import window_io
    win, scans = _scan_stream(n_scans=7, pts=2500, seed=synth.SEED_BASE + 74)
    o = odometry.frontend_opts(max(len(s) for s in scans), 1 << 16)
    path = str(tmp_path / "odo.bin")
    window_io.write_odometry_stream(path, o, scans, scan_match_cnt=2)
    poses, rows, info = window_io.run_demo_odometry(path)
    ctx = capi.Context(o)
    od = odometry.ScanToMapOdometry(ctx, scan_match_cnt=2)
    for k, sc in enumerate(scans):
        p, rounds = od.run(sc, max_points=o.max_points_per_scan)
        assert np.array_equal(p, poses[k]), (k, np.abs(p - poses[k]).max())
        assert rows[k]["rounds"] == len(rounds) == (0 if k == 0 else (8 if k == 1 else 2))
        assert rows[k]["kept"] == sum(kk for _, kk in rounds) and rows[k]["iterations"] == sum(int(s.iterations) for s, _ in rounds)
        if k >= 1:
            assert rows[k]["map_points"] == od.map_points
    assert info["scans"] == len(scans) and info["ms_per_scan"] > 0
    ctx.close()
