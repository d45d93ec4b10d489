// mini_eigen_probe.cpp -- test infrastructure: the linear-algebra stand-in the reference's factor layer is compiled against (include/mini_eigen.hpp:
// LLT, SelfAdjointEigenSolver, inverse, quaternion algebra) exposed through a C interface, so that tests/test_mini_eigen.py can hold it to numpy on
// random inputs.  The stand-in is load-bearing for every "reference-pinned" parity number (oracle/_ref), hence its own property tests.
// Build: make -C oracle/ref_shim probe      (no reference sources involved)
#include <Eigen/Dense>
#include <Eigen/Geometry>

using Eigen::MatrixXd;
typedef Eigen::Matrix<double, Eigen::Dynamic, Eigen::Dynamic, Eigen::RowMajor> RowMat;

extern "C" {
// lower Cholesky factor of a row-major SPD matrix; returns 1 when the factorisation reports success
int me_llt(int n, const double* a, double* l_out) {
    MatrixXd A(n, n);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) A(i, j) = a[i * n + j];
    Eigen::LLT<MatrixXd> llt(A);
    const MatrixXd L = llt.matrixL();
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) l_out[i * n + j] = L(i, j);
    return llt.info() == Eigen::Success ? 1 : 0;
}
// eigenvalues (ascending, as Eigen orders them) and eigenvectors (columns, row-major out) of a symmetric matrix
void me_eigh(int n, const double* a, double* w_out, double* v_out) {
    MatrixXd A(n, n);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) A(i, j) = a[i * n + j];
    Eigen::SelfAdjointEigenSolver<MatrixXd> es(A);
    for (int i = 0; i < n; ++i) w_out[i] = es.eigenvalues()(i);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) v_out[i * n + j] = es.eigenvectors()(i, j);
}
void me_inverse(int n, const double* a, double* inv_out) {
    MatrixXd A(n, n);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) A(i, j) = a[i * n + j];
    const MatrixXd B = A.inverse();
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) inv_out[i * n + j] = B(i, j);
}
// quaternions as (w, x, y, z): product, rotation of a vector, rotation matrix (row-major), inverse, normalised
void me_quat(const double* q1, const double* q2, const double* v, double* prod, double* rotated, double* R, double* inv, double* unit) {
    const Eigen::Quaterniond a(q1[0], q1[1], q1[2], q1[3]), b(q2[0], q2[1], q2[2], q2[3]);
    const Eigen::Quaterniond p = a * b;
    prod[0] = p.w(); prod[1] = p.x(); prod[2] = p.y(); prod[3] = p.z();
    const Eigen::Vector3d r = a * Eigen::Vector3d(v[0], v[1], v[2]);
    rotated[0] = r(0); rotated[1] = r(1); rotated[2] = r(2);
    const Eigen::Matrix3d M = a.toRotationMatrix();
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R[3 * i + j] = M(i, j);
    const Eigen::Quaterniond ai = a.inverse(), au = a.normalized();
    inv[0] = ai.w(); inv[1] = ai.x(); inv[2] = ai.y(); inv[3] = ai.z();
    unit[0] = au.w(); unit[1] = au.x(); unit[2] = au.y(); unit[3] = au.z();
}
}
