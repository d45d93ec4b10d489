"""ctypes binding of libglio_hip.so (include/glio_hip.h) and the host-side mirror of the reference's
sliding-window call sequence.  There is no CPU fallback: if the library or a HIP device is missing,
construction raises."""
import ctypes as C
import os

import numpy as np

from . import ctypes_types as T
from . import synth

_LIB = None
LIB_PATH = os.environ.get("GLIO_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libglio_hip.so")

(KERNEL_LIDAR_LINEARIZE, KERNEL_FULL_LINEARIZE, KERNEL_TR_STEP, KERNEL_ASSOCIATE, KERNEL_MAP_BUILD, KERNEL_MARGINALIZE, KERNEL_STREAM_READ,
 KERNEL_LINEARIZE_ALL, KERNEL_TR_STEP_STEADY) = range(9)
LIDAR_F64, LIDAR_F32_MFMA = 0, 1


class GlioError(RuntimeError):
    pass


def load():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise GlioError(f"{LIB_PATH} is missing: build it with `python -m glio_amd.build` (hipcc, gfx950). "
                            "There is no CPU fallback for the GLIO hot path.")
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 under the same SONAME as /opt/rocm's.
        # Whichever is loaded first serves both; torch cannot start on the system one ("No HIP GPUs are available"), so
        # when torch is installed it goes first.  (A C++ caller without torch simply uses /opt/rocm's.)
        if os.environ.get("GLIO_NO_TORCH_PRELOAD") != "1":
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        lib = C.CDLL(LIB_PATH)
        lib.glio_last_error.restype = C.c_char_p
        lib.glio_destroy.restype = None
        lib.glio_opts_default.restype = None
        _LIB = lib
    return _LIB


def _check(rc):
    if rc != 0:
        raise GlioError(f"libglio_hip error {rc}: {load().glio_last_error().decode()}")


def device_count():
    return load().glio_device_count()


# pcl::PointXYZI as numpy sees it (GLIO/include/utils/common.h: PointType): 32-byte records, x y z at 0, intensity at byte 16
PCL_XYZI = np.dtype({"names": ["x", "y", "z", "intensity"], "formats": ["<f4", "<f4", "<f4", "<f4"], "offsets": [0, 4, 8, 16], "itemsize": 32})
PCL_XYZI_INTENSITY_OFFSET = 16


def to_pcl_xyzi(xyzi):
    """[n][4] float32 -> the 32-byte records a pcl::PointCloud<pcl::PointXYZI> holds (padding filled with garbage on purpose: it must not matter)"""
    xyzi = np.asarray(xyzi, np.float32)
    out = np.frombuffer(np.random.default_rng(7).integers(0, 255, len(xyzi) * 32, dtype=np.uint8).tobytes(), dtype=PCL_XYZI).copy()
    out["x"], out["y"], out["z"], out["intensity"] = xyzi[:, 0], xyzi[:, 1], xyzi[:, 2], xyzi[:, 3]
    return out


class Context:
    """One glio_ctx = the device-resident state of one sliding window (one HIP stream)."""

    def __init__(self, opts, device=0):
        lib = load()
        if lib.glio_device_count() < 1:
            raise GlioError("no HIP device visible: libglio_hip has no CPU fallback")
        self.opts = opts
        self.W = opts.window
        self._h = C.c_void_p()
        _check(lib.glio_create(device, C.byref(opts), C.byref(self._h)))
        self._keep = []

    def close(self):
        if self._h:
            load().glio_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- uploads
    def set_map(self, map_pts):
        _check(load().glio_set_map(self._h, T.fptr(map_pts), len(map_pts)))

    # ---- strided point input: clouds as records of `stride` bytes (x y z floats at 0, intensity float at `ioff`); pcl::PointXYZI = PCL_XYZI
    @staticmethod
    def _raw(points):
        pts = np.ascontiguousarray(points)
        return pts, pts.ctypes.data_as(C.c_void_p), len(pts), pts.dtype.itemsize

    def set_map_strided(self, points, ioff):
        pts, ptr, n, stride = self._raw(points)
        _check(load().glio_set_map_strided(self._h, ptr, n, stride, ioff))

    def set_scan_strided(self, slot, points, ioff):
        pts, ptr, n, stride = self._raw(points)
        _check(load().glio_set_scan_strided(self._h, slot, ptr, n, stride, ioff))

    def localmap_push_strided(self, points, ioff, q, t):
        pts, ptr, n, stride = self._raw(points)
        q = np.ascontiguousarray(q, float); t = np.ascontiguousarray(t, float)
        _check(load().glio_localmap_push_strided(self._h, ptr if n else None, n, stride, ioff, T.dptr(q), T.dptr(t)))

    # ---- device-resident local map
    def localmap_config(self, width, leaf, max_points_per_keyframe):
        _check(load().glio_localmap_config(self._h, width, C.c_float(leaf), max_points_per_keyframe))

    def localmap_push(self, cloud, q, t):
        cloud = np.ascontiguousarray(cloud, np.float32)
        q = np.ascontiguousarray(q, float); t = np.ascontiguousarray(t, float)
        _check(load().glio_localmap_push(self._h, T.fptr(cloud) if len(cloud) else None, len(cloud), T.dptr(q), T.dptr(t)))

    def localmap_push_scan(self, scan_slot, lidar_offset, q, t):
        """push the scan that set_scan put into window slot `scan_slot` (no second upload): body point = scan point - lidar_offset (float)"""
        off = np.ascontiguousarray(lidar_offset, np.float32); q = np.ascontiguousarray(q, np.float64); t = np.ascontiguousarray(t, np.float64)
        _check(load().glio_localmap_push_scan(self._h, int(scan_slot), T.fptr(off), T.dptr(q), T.dptr(t)))

    def localmap_set_accumulation(self, mode):
        """0: exact fixed-point voxel sums (default); 1: pcl::VoxelGrid's float sums in concatenation order (bit-identical to the oracle's map)"""
        _check(load().glio_localmap_set_accumulation(self._h, int(mode)))

    def localmap_build(self):
        n = C.c_int()
        _check(load().glio_localmap_build(self._h, C.byref(n)))
        return n.value

    def localmap_read(self):
        n = C.c_int()
        _check(load().glio_localmap_read(self._h, None, 0, C.byref(n)))
        out = np.zeros((max(n.value, 1), 4), np.float32)
        _check(load().glio_localmap_read(self._h, T.fptr(out), n.value, C.byref(n)))
        return out[:n.value]

    def set_scan(self, slot, scan):
        _check(load().glio_set_scan(self._h, slot, T.fptr(scan), len(scan)))

    def localmap_push_scan_ahead_and_build(self, lidar_offset, q, t):
        """behind set_scan_ahead: the next call's local map (the cloud just sent, at the new keyframe's pose) on the upload stream; returns the map size"""
        off = np.ascontiguousarray(lidar_offset, np.float32); q = np.ascontiguousarray(q, float); t = np.ascontiguousarray(t, float)
        n = C.c_int()
        _check(load().glio_localmap_push_scan_ahead_and_build(self._h, T.fptr(off), T.dptr(q), T.dptr(t), C.byref(n)))
        return n.value

    def set_scan_ahead(self, scan):
        """the NEXT keyframe's scan into the ring row that is slot W - 1 after the next slide_window() (the current slot 0's scan is gone afterwards)"""
        _check(load().glio_set_scan_ahead(self._h, T.fptr(scan), len(scan)))

    def associate(self, slot, scan, q, t):
        cnt = C.c_int()
        q = np.ascontiguousarray(q, float); t = np.ascontiguousarray(t, float)
        _check(load().glio_associate(self._h, slot, T.fptr(scan), len(scan), T.dptr(q), T.dptr(t), C.byref(cnt)))
        return cnt.value

    def associate_resident(self, slot, q, t):
        cnt = C.c_int()
        q = np.ascontiguousarray(q, float); t = np.ascontiguousarray(t, float)
        _check(load().glio_associate_resident(self._h, slot, T.dptr(q), T.dptr(t), C.byref(cnt)))
        return cnt.value

    def select_correspondences_window(self, per_slot):
        """per_slot[s]: None (slot untouched) or the indices slot s keeps, in order -- featureSelection of the whole window in one call"""
        W = self.W
        offs = np.zeros(W + 1, np.int32); changed = np.zeros(W, np.uint8); parts = []
        for s in range(W):
            if per_slot[s] is not None:
                changed[s] = 1; parts.append(np.ascontiguousarray(per_slot[s], np.int32))
            offs[s + 1] = offs[s] + (len(per_slot[s]) if per_slot[s] is not None else 0)
        idx = np.concatenate(parts).astype(np.int32) if parts else np.zeros(0, np.int32)
        _check(load().glio_select_correspondences_window(self._h, T.iptr(offs), T.iptr(idx) if len(idx) else None, changed.ctypes.data_as(C.POINTER(C.c_uint8))))

    def select_correspondences(self, slot, indices):
        idx = np.ascontiguousarray(indices, np.int32)
        _check(load().glio_select_correspondences(self._h, slot, T.iptr(idx) if len(idx) else None, len(idx)))

    def slide_window(self):
        _check(load().glio_slide_window(self._h))

    def associate_window_async(self, quats, trans):
        """enqueue the association of all W slots and return at once (counts: associate_window_counts, or implicitly at the next solve)"""
        q = np.ascontiguousarray(quats, np.float64); t = np.ascontiguousarray(trans, np.float64)
        _check(load().glio_associate_window_async(self._h, T.dptr(q), T.dptr(t)))

    def associate_window_counts(self):
        out = np.zeros(self.W, np.int32)
        _check(load().glio_associate_window_counts(self._h, T.iptr(out)))
        return out

    def associate_window(self, quats, trans):
        quats = np.ascontiguousarray(quats, float); trans = np.ascontiguousarray(trans, float)
        cnt = np.zeros(self.W, np.int32)
        _check(load().glio_associate_window(self._h, T.dptr(quats), T.dptr(trans), T.iptr(cnt)))
        return cnt

    def set_correspondences(self, slot, pts, planes, scores):
        pts = np.ascontiguousarray(pts, np.float32); planes = np.ascontiguousarray(planes, np.float32)
        scores = np.ascontiguousarray(scores, np.float64)
        n = len(scores)
        if n == 0:
            pts = np.zeros((1, 4), np.float32); planes = np.zeros((1, 4), np.float32); scores = np.zeros(1)
        _check(load().glio_set_correspondences(self._h, slot, T.fptr(pts), T.fptr(planes), T.dptr(scores), n))

    def get_correspondences(self, slot):
        cap = self.opts.max_points_per_scan
        pts = np.zeros((cap, 4), np.float32); planes = np.zeros((cap, 4), np.float32); scores = np.zeros(cap)
        cnt = C.c_int()
        _check(load().glio_get_correspondences(self._h, slot, T.fptr(pts), T.fptr(planes), T.dptr(scores), cap, C.byref(cnt)))
        n = cnt.value
        return pts[:n].copy(), planes[:n].copy(), scores[:n].copy()

    @staticmethod
    def marshal_imu(preints, slots=None):
        """The C arrays glio_set_imu takes (a C++ caller owns these natively; Python pays ~0.5 ms to build them)."""
        n = len(preints)
        arr = T.preint_array(n)
        for k, p in enumerate(preints):
            synth.fill_preint(arr[k], p)
        slots = np.arange(max(n, 1), dtype=np.int32) if slots is None else np.ascontiguousarray(slots, np.int32)
        return n, arr, slots

    @staticmethod
    def marshal_gnss(frame, dd, dop):
        return frame, len(dd), (T.GlioDdPsr * max(len(dd), 1))(*dd), len(dop), (T.GlioDoppler * max(len(dop), 1))(*dop)

    def set_imu_marshalled(self, m):
        n, arr, slots = m
        _check(load().glio_set_imu(self._h, n, arr, T.iptr(slots)))

    def set_gnss_marshalled(self, m):
        frame, ndd, dd_arr, ndop, dop_arr = m
        _check(load().glio_set_gnss(self._h, C.byref(frame) if frame is not None else None, ndd, dd_arr, ndop, dop_arr))

    def set_imu(self, preints, slots=None):
        self.set_imu_marshalled(self.marshal_imu(preints, slots))

    def set_prior(self, prior):
        ps = synth.prior_struct(prior)
        self._keep = [prior]          # glio_set_prior copies synchronously; only the latest is held (for the caller's convenience)
        _check(load().glio_set_prior(self._h, C.byref(ps)))

    def set_gnss(self, frame, dd, dop):
        self.set_gnss_marshalled(self.marshal_gnss(frame, dd, dop))

    def load_window(self, win, corr=None, use_gnss=True, use_prior=True, use_imu=True):
        """Upload a synth.Window: correspondences (pre-made, parity hook) + all small factors."""
        if corr is not None:
            for s in range(win.W):
                self.set_correspondences(s, *corr[s])
        self.set_imu(win.preints if use_imu else [])
        self.set_prior(win.prior if use_prior else None)
        if use_gnss and win.frame is not None:
            self.set_gnss(win.frame, win.dd, win.dop)
        else:
            self.set_gnss(None, [], [])

    # ---- compute
    def linearize(self, state, want_H=True):
        n = 15 * self.W + state.n_ddt
        H = np.zeros((n, n)) if want_H else None
        g = np.zeros(n) if want_H else None
        cost = C.c_double()
        cs = state.c()
        _check(load().glio_linearize(self._h, C.byref(cs), T.dptr(H) if want_H else None, T.dptr(g) if want_H else None, C.byref(cost)))
        return H, g, cost.value

    def solve(self, state):
        s = state.copy()
        cs = s.c()
        summ = T.GlioSummary()
        _check(load().glio_solve(self._h, C.byref(cs), C.byref(summ)))
        return s, summ

    def marginalize(self, state):
        """Marginalize slot 0 at `state` (Estimator.cpp:2462-2607): returns the glio_prior fields of the next window."""
        W = self.W
        n = 6 * (W - 1) + 9
        nb = 2 * (W - 1) + 1
        out = dict(n=n, lin_jac=np.zeros((n, n)), lin_res=np.zeros(n), blk_slot=np.zeros(nb, np.int32),
                   blk_kind=np.zeros(nb, np.int32), blk_idx=np.zeros(nb, np.int32), blk_x0=np.zeros((nb, 9)))
        cs = state.c()
        on, onb = C.c_int32(), C.c_int32()
        _check(load().glio_marginalize(self._h, C.byref(cs), T.dptr(out["lin_jac"]), T.dptr(out["lin_res"]), T.iptr(out["blk_slot"]),
                                       T.iptr(out["blk_kind"]), T.iptr(out["blk_idx"]), T.dptr(out["blk_x0"]), C.byref(on), C.byref(onb)))
        assert on.value == n and onb.value == nb
        return out

    def marginalize_keep(self, state):
        cs = state.c()
        _check(load().glio_marginalize_keep(self._h, C.byref(cs)))

    def marginalize_keep_async(self, state):
        """enqueue the marginalization and return; marginalize_keep_finish() (or the next entry point that reads the prior) waits"""
        cs = state.c()
        _check(load().glio_marginalize_keep_async(self._h, C.byref(cs)))

    def marginalize_keep_finish(self):
        _check(load().glio_marginalize_keep_finish(self._h))

    def time_kernel(self, which, reps=20):
        ms = C.c_float()
        _check(load().glio_time_kernel(self._h, which, reps, C.byref(ms)))
        return ms.value

    def time_solve(self, state, reps=5):
        ms = C.c_float()
        summ = T.GlioSummary()
        cs = state.c()
        _check(load().glio_time_solve(self._h, C.byref(cs), reps, C.byref(ms), C.byref(summ)))
        return ms.value, summ

    def set_stream(self, stream_ptr):
        _check(load().glio_set_stream(self._h, C.c_void_p(stream_ptr)))

    # ---- single-factor evaluators (Ceres Evaluate convention)
    def eval_lidar_plane(self, cp, plane, score, t, q):
        cp = np.ascontiguousarray(cp, np.float32); plane = np.ascontiguousarray(plane, np.float32)
        r = np.zeros(1); Jt = np.zeros(3); Jq = np.zeros(4)
        params = [np.ascontiguousarray(t, float), np.ascontiguousarray(q, float)]
        P = (T.c_double_p * 2)(T.dptr(params[0]), T.dptr(params[1]))
        J = (T.c_double_p * 2)(T.dptr(Jt), T.dptr(Jq))
        _check(load().glio_eval_lidar_plane(self._h, T.fptr(cp), T.fptr(plane), C.c_double(score), P, T.dptr(r), J))
        return r[0], Jt, Jq

    def eval_imu(self, pre, params):
        ps = T.GlioPreint()
        synth.fill_preint(ps, pre)
        params = [np.ascontiguousarray(p, float) for p in params]
        P = (T.c_double_p * 6)(*[T.dptr(p) for p in params])
        r = np.zeros(15)
        Js = [np.zeros((15, s)) for s in (3, 4, 9, 3, 4, 9)]
        J = (T.c_double_p * 6)(*[T.dptr(j) for j in Js])
        _check(load().glio_eval_imu(self._h, C.byref(ps), P, T.dptr(r), J))
        return r, Js


def _ptrs(arrs):
    return (T.c_double_p * len(arrs))(*[T.dptr(a) if a is not None else None for a in arrs])


def _ctx_eval_methods():
    def eval_dd_psr(self, f, Pi, Pj, yaw, anc):
        params = [np.ascontiguousarray(Pi, float), np.ascontiguousarray(Pj, float), np.array([yaw], float), np.ascontiguousarray(anc, float)]
        r = np.zeros(19); Js = [np.zeros((19, 3)), np.zeros((19, 3)), None, None]
        _check(load().glio_eval_dd_psr(self._h, C.byref(f), _ptrs(params), T.dptr(r), _ptrs(Js)))
        return r, Js[:2]

    def eval_doppler(self, f, Pi, SBi, Pj, SBj, ddt, yaw, anc):
        params = [np.ascontiguousarray(a, float) for a in (Pi, SBi, Pj, SBj, ddt)] + [np.array([yaw], float), np.ascontiguousarray(anc, float)]
        r = np.zeros(1); Js = [np.zeros(3), np.zeros(9), np.zeros(3), np.zeros(9), np.zeros(1), None, None]
        _check(load().glio_eval_doppler(self._h, C.byref(f), _ptrs(params), T.dptr(r), _ptrs(Js)))
        return r[0], Js[:5]

    def eval_marginalization(self, prior, params):
        ps = synth.prior_struct(prior)
        n = prior["n"]
        sizes = [3 if k == 0 else (4 if k == 1 else 9) for k in prior["blk_kind"]]
        params = [np.ascontiguousarray(p, float) for p in params]
        r = np.zeros(n); Js = [np.zeros((n, s)) for s in sizes]
        _check(load().glio_eval_marginalization(self._h, C.byref(ps), _ptrs(params), T.dptr(r), _ptrs(Js)))
        return r, Js

    def eval_binary_plane(self, cp, pnc, score, t1, q1, t2, q2):
        cp = np.ascontiguousarray(cp, np.float32); pnc = np.ascontiguousarray(pnc, float)
        params = [np.ascontiguousarray(a, float) for a in (t1, q1, t2, q2)]
        r = np.zeros(1); Js = [np.zeros(3), np.zeros(4), np.zeros(3), np.zeros(4)]
        _check(load().glio_eval_binary_plane(self._h, T.fptr(cp), T.dptr(pnc), C.c_double(score), _ptrs(params), T.dptr(r), _ptrs(Js)))
        return r[0], Js
    for fn in (eval_dd_psr, eval_doppler, eval_marginalization, eval_binary_plane):
        setattr(Context, fn.__name__, fn)


_ctx_eval_methods()


def lidar_pose(opts, q, t):
    """Q2 = Q * q_lb^-1, T2 = T - Q2 * t_lb: the LiDAR pose handed to findCorrespondingSurfFeatures
    (reference GLIO/src/Estimator.cpp:2216-2217)."""
    qlb = np.array(opts.q_lb)
    q2 = synth.qmul(np.asarray(q, float), synth.qconj(qlb) / (qlb @ qlb))
    t2 = np.asarray(t, float) - synth.q2R(q2 / np.linalg.norm(q2)) @ np.array(opts.t_lb)
    return q2, t2
