"""Property tests of the Eigen stand-in (oracle/ref_shim/include/mini_eigen.hpp) against numpy.

The reference's factor layer (GLIO/include/factors/*.h, MarginalizationFactor.cpp) is compiled UNMODIFIED into oracle/_ref, but against this stand-in,
not against Eigen 3.3.3 (absent from the image): the reference's FORMULAS are the reference's, the linear algebra under them is ours.  Everything the
factors take from it -- LLT (ImuFactor.h:44-45 sqrt_info), SelfAdjointEigenSolver (MarginalizationFactor.cpp:176-201), inverse (ImuFactor.h:44,
Preintegration.h), the quaternion product / rotation / rotation matrix / inverse / normalisation -- is held here to numpy's LAPACK-backed results on
1000 random inputs each.  CPU only; builds the probe with g++ (no reference tree needed)."""
import ctypes as C

import numpy as np
import pytest

from oracle import pyref

N = 1000


@pytest.fixture(scope="module")
def me():
    lib = C.CDLL(pyref.build_probe())
    return lib


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _spd(rng, n):
    A = rng.normal(0, 1, (n, n))
    return A @ A.T + n * np.diag(rng.uniform(0.1, 1.0, n))


@pytest.mark.parametrize("n", [3, 6, 15])
def test_llt_lower_factor(me, n):
    rng = np.random.default_rng(100 + n)
    for _ in range(N // 3):
        A = _spd(rng, n)
        L = np.zeros((n, n))
        assert me.me_llt(n, _p(np.ascontiguousarray(A)), _p(L)) == 1
        Lw = np.linalg.cholesky(A)
        assert np.allclose(L, Lw, rtol=1e-11, atol=1e-12) and not np.triu(L, 1).any()
        assert np.allclose(L @ L.T, A, rtol=1e-12, atol=1e-12)
    bad = -np.eye(n)
    assert me.me_llt(n, _p(np.ascontiguousarray(bad)), _p(np.zeros((n, n)))) == 0          # not positive definite: reported


@pytest.mark.parametrize("n", [3, 15, 24])
def test_self_adjoint_eigen_solver(me, n):
    """eigenvalues ascending like Eigen's; A V = V diag(w), V orthonormal.  Includes the rank-deficient case the marginalization feeds it
    (MarginalizationFactor.cpp:186: eigenvalues below eps are zeroed)."""
    rng = np.random.default_rng(200 + n)
    for k in range(N // 4):
        A = _spd(rng, n)
        if k % 5 == 0:                                 # rank n - 2
            Q = np.linalg.qr(rng.normal(0, 1, (n, n)))[0]
            w = np.r_[np.zeros(2), rng.uniform(0.5, 50.0, n - 2)]
            A = (Q * w) @ Q.T
            A = 0.5 * (A + A.T)
        w, V = np.zeros(n), np.zeros((n, n))
        me.me_eigh(n, _p(np.ascontiguousarray(A)), _p(w), _p(V))
        ww = np.linalg.eigvalsh(A)
        scale = max(1.0, abs(ww).max())
        assert np.all(np.diff(w) >= -1e-12 * scale) and np.allclose(w, ww, rtol=0, atol=1e-11 * scale)
        assert np.allclose(V.T @ V, np.eye(n), atol=1e-11) and np.allclose(A @ V, V * w, atol=1e-10 * scale)


@pytest.mark.parametrize("n", [3, 9, 15])
def test_inverse(me, n):
    rng = np.random.default_rng(300 + n)
    for k in range(N // 3):
        A = _spd(rng, n) if k % 2 else rng.normal(0, 1, (n, n)) + 3.0 * np.eye(n)          # SPD (covariances) and general well-conditioned
        B = np.zeros((n, n))
        me.me_inverse(n, _p(np.ascontiguousarray(A)), _p(B))
        assert np.allclose(B, np.linalg.inv(A), rtol=1e-9, atol=1e-11) and np.allclose(A @ B, np.eye(n), atol=1e-10)


def _qmul(a, b):
    w1, v1, w2, v2 = a[0], a[1:], b[0], b[1:]
    return np.r_[w1 * w2 - v1 @ v2, w1 * v2 + w2 * v1 + np.cross(v1, v2)]


def test_quaternion_algebra(me):
    rng = np.random.default_rng(400)
    for k in range(N):
        q1, q2, v = rng.normal(0, 1, 4), rng.normal(0, 1, 4), rng.normal(0, 3, 3)
        if k % 2 == 0:
            q1 /= np.linalg.norm(q1); q2 /= np.linalg.norm(q2)
        prod, rot, R, inv, unit = np.zeros(4), np.zeros(3), np.zeros((3, 3)), np.zeros(4), np.zeros(4)
        me.me_quat(_p(q1), _p(q2), _p(v), _p(prod), _p(rot), _p(R), _p(inv), _p(unit))
        assert np.allclose(prod, _qmul(q1, q2), atol=1e-13)
        assert np.allclose(inv, np.r_[q1[0], -q1[1:]] / (q1 @ q1), atol=1e-13) and np.allclose(unit, q1 / np.linalg.norm(q1), atol=1e-14)
        if k % 2 == 0:                                 # unit quaternions: q v q^-1, and the rotation matrix is that map
            want = _qmul(_qmul(q1, np.r_[0.0, v]), np.r_[q1[0], -q1[1:]])[1:]
            assert np.allclose(rot, want, atol=1e-12) and np.allclose(R @ v, want, atol=1e-12)
            assert np.allclose(R.T @ R, np.eye(3), atol=1e-13) and np.isclose(np.linalg.det(R), 1.0, atol=1e-13)
