"""Generates tests/golden/next_rows_small.npz: frozen ORACLE answers (not reference output -- see make_golden.py) for the rows
SURVEY section 8f ranks next: one keyframe pair of the batch association (f2), two rounds of the front-end scan-to-map
odometry (f3) and the voxel-grid local map of a three-keyframe ring (f4).

    python tests/golden/make_golden_next.py
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))


def make_inputs():
    from glio_amd import synth
    win = synth.make_window(W=3, pts_per_scan=1500, seed=synth.SEED_BASE + 91, perturb=(0.03, 0.2, 0.0), scan_radius=14.0, map_density=1.0)
    tlb = np.array(win.opts.t_lb, np.float32)
    body = []
    for s in range(win.W):
        c = win.scans[s].copy(); c[:, :3] -= tlb
        body.append(np.ascontiguousarray(c))
    return win, body


def digest(win, body):
    h = hashlib.sha256()
    for a in [win.map_pts] + body + [win.init.trans, win.init.quat, win.gt.trans, win.gt.quat]:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


class OracleBackend:
    """The odometry loop of glio_amd/odometry.py on the CPU oracle (same duck type as capi.Context)."""

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
        from glio_amd import synth
        win = synth.Window(opts=self.opts, W=1, gt=None, init=None, kf_times=None, scans=None, scan_plane_id=None, map_pts=self.map, scene=None)
        return self.po.Problem(win, self.corr, use_gnss=False, use_prior=False, use_imu=False).solve(state)


def oracle_outputs(win, body):
    from glio_amd import odometry
    from oracle import pyoracle as po
    out = {}
    poses = np.c_[win.init.trans, win.init.quat]
    cp, nc, sc, _ = po.associate_pair(body[0], poses[0], body[1], poses[1])                 # f2
    out.update(pair_count=np.array(len(sc)), pair_cp=cp[:16], pair_nc=nc[:16], pair_score=sc[:16])
    o = odometry.frontend_opts(len(body[0]), len(win.map_pts))                             # f3
    odo = odometry.ScanToMapOdometry(OracleBackend(o))
    odo.set_map(win.map_pts)
    pose, rounds = odo.update(body[0], np.r_[win.init.quat[0], win.init.trans[0]], match_cnt=2)
    out.update(odo_pose=pose, odo_iterations=np.array([r[0].iterations for r in rounds]), odo_kept=np.array([r[1] for r in rounds]))
    glob = [po.transform_cloud(c, win.gt.quat[s], win.gt.trans[s]) for s, c in enumerate(body)]   # f4
    vox, _ = po.voxel_grid(np.vstack(glob), 0.4)
    out.update(map_count=np.array(len(vox)), map_head=vox[:32])
    return out


if __name__ == "__main__":
    win, body = make_inputs()
    out = oracle_outputs(win, body)
    out["input_sha256"] = np.array(digest(win, body))
    path = os.path.join(HERE, "next_rows_small.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; pair kept", int(out["pair_count"]), "odometry iterations", out["odo_iterations"], "voxels", int(out["map_count"]))
