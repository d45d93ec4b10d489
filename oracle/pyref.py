"""ctypes binding of oracle/_ref/libglio_ref.so: the REFERENCE'S OWN factor classes (GLIO/include/factors/*.h, math_tools.h,
GLIO/src/MarginalizationFactor.cpp, gnss_comm/src/gnss_utility.cpp) compiled unmodified from /root/reference against the stand-in
headers of oracle/ref_shim/include (recipe: oracle/ref_shim/Makefile).

TEST INFRASTRUCTURE ONLY -- it validates the restatement in oracle/*.c (tests/test_oracle_ref.py).  Nothing under glio_amd/ may
import this module or load the library (tests/test_abi.py guards that).  The library can only be BUILT where /root/reference
exists (this container); a prebuilt oracle/_ref/ travels to the GPU box with the tree, where it is not needed by any test.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from glio_amd import ctypes_types as T

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_ref", "libglio_ref.so")
REFERENCE = os.environ.get("GLIO_REFERENCE_ROOT", "/root/reference")


def available():
    return os.path.exists(_SO) or os.path.isdir(os.path.join(REFERENCE, "GLIO", "include", "factors"))


def build(force=False):
    """make decides what is stale; without the reference tree a prebuilt library is used as it is."""
    if os.path.isdir(os.path.join(REFERENCE, "GLIO", "include", "factors")):
        subprocess.check_call(["make", "-C", os.path.join(_HERE, "ref_shim"), "-j3", "REF=" + REFERENCE] + (["-B"] if force else []), stdout=subprocess.DEVNULL)
    if not os.path.exists(_SO):
        raise FileNotFoundError("oracle/_ref/libglio_ref.so: no reference tree at %s and no prebuilt library" % REFERENCE)
    return _SO


def build_probe():
    """oracle/_ref/libme_probe.so: the Eigen stand-in behind a C interface (tests/test_mini_eigen.py); needs no reference tree"""
    subprocess.check_call(["make", "-C", os.path.join(_HERE, "ref_shim"), "probe"], stdout=subprocess.DEVNULL)
    return os.path.join(_HERE, "_ref", "libme_probe.so")


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.ref_marginalize.restype = C.c_int
        _lib.ref_set_param.argtypes = [C.c_char_p, C.c_double]
        _lib.ref_set_param.restype = None
        _lib.ref_ecef2rotation.restype = None
    return _lib


def _pp(arrs):
    P = (T.c_double_p * len(arrs))()
    for i, a in enumerate(arrs):
        P[i] = T.dptr(a) if a is not None else None
    return P


def _d(a):
    return np.ascontiguousarray(a, np.float64)


def set_param(name, value):
    lib().ref_set_param(name.encode(), float(value))


def _eval(fn, head_args, params, nres, sizes, want_J=True):
    r = np.zeros(nres)
    J = [np.zeros((nres, s)) for s in sizes]
    params = [_d(p) for p in params]
    rc = fn(*head_args, _pp(params), T.dptr(r), _pp(J) if want_J else None)
    assert rc == 0, rc
    return r, J


def eval_lidar_plane(cp, n, d, score, qlb, tlb, t, q, want_J=True):
    """LidarPlaneNormFactor::Create(...)->Evaluate: residual[1], jacobians (1x3 wrt t, 1x4 wrt q)"""
    cp, n, qlb, tlb = _d(cp), _d(n), _d(qlb), _d(tlb)
    return _eval(lib().ref_eval_lidar_plane, (T.dptr(cp), T.dptr(n), C.c_double(d), C.c_double(score), T.dptr(qlb), T.dptr(tlb)), [t, q], 1, [3, 4], want_J)


def eval_binary_plane(cp, pnc, score, t1, q1, t2, q2, want_J=True):
    cp, pnc = _d(cp), _d(pnc)
    return _eval(lib().ref_eval_binary_plane, (T.dptr(cp), T.dptr(pnc), C.c_double(score)), [t1, q1, t2, q2], 1, [3, 4, 3, 4], want_J)


def eval_plane_incre(cp, n, d, q, t, want_J=True):
    cp, n = _d(cp), _d(n)
    return _eval(lib().ref_eval_plane_incre, (T.dptr(cp), T.dptr(n), C.c_double(d)), [q, t], 1, [4, 3], want_J)


def eval_delta_q(dq, qi, qj, want_J=True):
    dq = _d(dq)
    return _eval(lib().ref_eval_delta_q, (T.dptr(dq),), [qi, qj], 3, [4, 4], want_J)


def eval_relative_pose(dq, dp, p1, q1, p2, q2, want_J=True):
    dq, dp = _d(dq), _d(dp)
    return _eval(lib().ref_eval_relative_pose, (T.dptr(dq), T.dptr(dp)), [p1, q1, p2, q2], 6, [3, 4, 3, 4], want_J)


def eval_imu(pre_struct, gravity, params, want_J=True):
    return _eval(lib().ref_eval_imu, (C.byref(pre_struct), C.c_double(gravity)), params, 15, [3, 4, 9, 3, 4, 9], want_J)


def preintegrate(acc0, gyr0, ba, bg, dt, acc, gyr):
    """Preintegration(acc0, gyr0, ba, bg) + push_back over the samples: a filled GlioPreint"""
    out = T.GlioPreint()
    dt, acc, gyr = _d(dt), _d(acc), _d(gyr)
    rc = lib().ref_preintegrate(T.dptr(_d(acc0)), T.dptr(_d(gyr0)), T.dptr(_d(ba)), T.dptr(_d(bg)), len(dt), T.dptr(dt), T.dptr(acc), T.dptr(gyr), C.byref(out))
    assert rc == 0
    return out


def eval_dd_psr(f, Pi, Pj, yaw, anc, want_J=True):
    r, J = _eval(lib().ref_eval_dd_psr, (C.byref(f),), [Pi, Pj, np.array([yaw], float), anc], 19, [3, 3, 1, 3], want_J)
    return r, J[:2]


def ddt_slots():
    return lib().ref_ddt_slots()


def eval_doppler(f, Pi, SBi, Pj, SBj, ddt, yaw, anc, want_J=True):
    n = ddt_slots()
    ddt_full = np.zeros(n)
    ddt_full[:len(ddt)] = ddt
    r, J = _eval(lib().ref_eval_doppler, (C.byref(f),), [Pi, SBi, Pj, SBj, ddt_full, np.array([yaw], float), anc], 1, [3, 9, 3, 9, n, 1, 3], want_J)
    return r[0], J


def ecef2rotation(ecef):
    R = np.zeros((3, 3))
    lib().ref_ecef2rotation(T.dptr(_d(ecef)), T.dptr(R))
    return R


def eval_marg(prior_dict, params, want_J=True):
    from glio_amd import synth
    ps = synth.prior_struct(prior_dict)
    n = prior_dict["n"]
    sizes = [3 if k == 0 else (4 if k == 1 else 9) for k in prior_dict["blk_kind"]]
    return _eval(lib().ref_eval_marg, (C.byref(ps),), params, n, sizes, want_J)


def marginalize(opts, state, offset, pts, planes, scores, imu01, prior_dict=None):
    """The reference's MarginalizationInfo over the estimator's factor list (Estimator.cpp:2462-2607).  Returns a prior dict in the
    REFERENCE'S block order (unordered_map iteration), slots already shifted."""
    from glio_amd import synth
    W = len(state.trans)
    n = 6 * (W - 1) + 9
    nbmax = 2 * (W - 1) + 1
    lin_jac, lin_res = np.zeros((n, n)), np.zeros(n)
    blk_slot, blk_kind, blk_idx = np.zeros(nbmax, np.int32), np.zeros(nbmax, np.int32), np.zeros(nbmax, np.int32)
    blk_x0 = np.zeros((nbmax, 9))
    nb = C.c_int32()
    ps = synth.prior_struct(prior_dict) if prior_dict is not None else None
    qlb, tlb = _d(list(opts.q_lb)), _d(list(opts.t_lb))
    tr, qu, sb = _d(state.trans), _d(state.quat), _d(state.speed_bias)
    offset = np.ascontiguousarray(offset, np.int32); pts = np.ascontiguousarray(pts, np.float32); planes = np.ascontiguousarray(planes, np.float32)
    scores = _d(scores)
    got = lib().ref_marginalize(W, T.dptr(tr), T.dptr(qu), T.dptr(sb), T.dptr(qlb), T.dptr(tlb), C.c_double(opts.huber_delta), C.c_double(opts.gravity),
                                T.iptr(offset), T.fptr(pts), T.fptr(planes), T.dptr(scores), C.byref(imu01), C.byref(ps) if ps is not None else None,
                                T.dptr(lin_jac), T.dptr(lin_res), T.iptr(blk_slot), T.iptr(blk_kind), T.iptr(blk_idx), T.dptr(blk_x0), C.byref(nb))
    assert got == n, (got, n)
    k = nb.value
    return dict(n=n, lin_jac=lin_jac, lin_res=lin_res, blk_slot=blk_slot[:k].copy(), blk_kind=blk_kind[:k].copy(), blk_idx=blk_idx[:k].copy(), blk_x0=blk_x0[:k].copy())
