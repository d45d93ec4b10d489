"""Strided point input (the wire-format half): every cloud-taking entry point accepts the 32-byte pcl::PointXYZI records of the reference's
PointCloud<PointType> (x y z at 0, intensity at byte 16, garbage in the padding) and produces the records of the packed float4 path, byte for byte."""
import numpy as np
import pytest

from glio_amd import batch, capi, synth

pytestmark = pytest.mark.gpu
IOFF = capi.PCL_XYZI_INTENSITY_OFFSET


def test_scan_map_and_local_map_from_pcl_records(small_window):
    win = small_window
    q2, t2 = capi.lidar_pose(win.opts, win.init.quat[0], win.init.trans[0])
    a = capi.Context(win.opts); a.set_map(win.map_pts)
    na = a.associate(0, win.scans[0], q2, t2); ra = [x.copy() for x in a.get_correspondences(0)]
    b = capi.Context(win.opts); b.set_map_strided(capi.to_pcl_xyzi(win.map_pts), IOFF)
    b.set_scan_strided(0, capi.to_pcl_xyzi(win.scans[0]), IOFF)
    nb = b.associate_resident(0, q2, t2); rb = b.get_correspondences(0)
    assert na == nb > 100 and all(np.array_equal(x.view(np.uint8), y.view(np.uint8)) for x, y in zip(ra, rb))
    # the device-resident local map: pushes of the same clouds, packed and strided, give the same voxel map
    for c in (a, b):
        c.localmap_config(5, 0.4, len(win.scans[0]))
    tlb = np.array(win.opts.t_lb, np.float32)
    for s in range(3):
        body = win.scans[s].copy(); body[:, :3] -= tlb
        a.localmap_push(body, win.gt.quat[s], win.gt.trans[s])
        b.localmap_push_strided(capi.to_pcl_xyzi(body), IOFF, win.gt.quat[s], win.gt.trans[s])
    assert a.localmap_build() == b.localmap_build() > 100
    assert np.array_equal(a.localmap_read().view(np.uint8), b.localmap_read().view(np.uint8))
    # layouts that cannot be a point are refused
    lib = capi.load()
    raw = capi.to_pcl_xyzi(win.scans[0])
    for stride, ioff in ((12, 12), (32, 8), (32, 30), (30, 16), (32, 32)):
        assert lib.glio_set_scan_strided(b._h, 0, raw.ctypes.data, len(raw), stride, ioff) != 0
    a.close(); b.close()


def test_batch_association_frames_from_pcl_records():
    win = synth.make_window(W=4, pts_per_scan=3000, seed=synth.SEED_BASE + 57, perturb=(0.03, 0.2, 0.0), scan_radius=14.0, map_density=1.0)
    tlb = np.array(win.opts.t_lb, np.float32)
    scans = []
    for s in range(win.W):
        c = win.scans[s].copy(); c[:, :3] -= tlb
        scans.append(np.ascontiguousarray(c))
    poses = np.c_[win.init.trans, win.init.quat]
    ci, cj = batch.pair_list(win.W, 1)
    out = []
    for strided in (False, True):
        ba = batch.BatchAssociation(win.W, 4096, 200000)
        for k in range(win.W):
            if strided:
                ba.set_frame_strided(k, capi.to_pcl_xyzi(scans[k]), IOFF)
            else:
                ba.set_frame(k, scans[k])
        cnt, tot = ba.run(poses, ci, cj)
        out.append((cnt.copy(), [x.copy() for x in ba.read()]))
        ba.close()
    assert np.array_equal(out[0][0], out[1][0]) and out[0][0].sum() > 1000
    for x, y in zip(out[0][1], out[1][1]):
        assert np.array_equal(x.view(np.uint8), y.view(np.uint8))
