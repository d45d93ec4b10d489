// mini_eigen.hpp -- TEST INFRASTRUCTURE (part of the oracle, never linked into the product).
//
// A small, EAGER (no expression templates) stand-in for the subset of Eigen 3.3 that the reference's factor
// layer uses, so that the reference's own sources
//     GLIO/include/utils/math_tools.h, GLIO/include/factors/*.h, GLIO/src/MarginalizationFactor.cpp,
//     gnss_comm/src/gnss_utility.cpp
// compile UNMODIFIED from /root/reference (recipe: oracle/ref_shim/Makefile -> oracle/_ref/libglio_ref.so).
// Eigen itself is not in this image (and not under /root/reference: README.md:79 names 3.3.3 as an external
// dependency).  What is restated here is Eigen's PUBLIC behaviour for the calls those files make:
// column-major storage by default, (w,x,y,z) quaternion constructor with (x,y,z,w) coefficient order,
// q * v as a rotation, q.inverse() = conjugate / squared norm, LLT lower factor, self-adjoint eigen
// decomposition with ascending eigenvalues, comma initialisation row by row, Map over caller memory with the
// storage order of its plain type.  Every operation evaluates into a plain Matrix at once: aliasing-safe by
// construction and the floating-point operation ORDER of a product is the textbook one (k ascending) -- it need
// not be Eigen's, the tests that use this library compare to 1e-12 relative, not bitwise.
#ifndef GLIO_ORACLE_MINI_EIGEN_HPP
#define GLIO_ORACLE_MINI_EIGEN_HPP

#include <cassert>
#include <cmath>
#include <cstddef>
#include <algorithm>
#include <cstring>
#include <iostream>
#include <numeric>
#include <limits>
#include <type_traits>
#include <utility>
#include <vector>

#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW
#define EIGEN_ALIGN16 __attribute__((aligned(16)))
#define EIGEN_WORLD_VERSION 3
#define EIGEN_MAJOR_VERSION 3
#define EIGEN_MINOR_VERSION 3

namespace Eigen {

const int Dynamic = -1;
enum StorageOptions { ColMajor = 0, RowMajor = 1, AutoAlign = 0, DontAlign = 2 };
enum { Lower = 1, Upper = 2 };
typedef std::ptrdiff_t Index;

template <class Derived> struct traits;
template <class S, int R, int C, int Opt = ColMajor, int MR = R, int MC = C> class Matrix;
template <class Xpr, int R, int C> class Block;
template <class Plain, int MapOpt = 0, class Stride = void> class Map;
template <class S> class Quaternion;
template <class S> class AngleAxis;
template <class S> class ArrayWrap;
template <class S> class DiagonalWrap;
template <class Derived> class CommaInitializer;

namespace internal {
template <int A, int B> struct pick_dim { static const int value = (A != Dynamic) ? A : B; };
template <class T> struct is_matrix_like : std::false_type {};
using std::sqrt; using std::abs;
}  // namespace internal

// ------------------------------------------------------------------------------------------------ MatrixBase
template <class Derived> class MatrixBase {
  public:
    typedef typename traits<Derived>::Scalar Scalar;
    enum { RowsAtCompileTime = traits<Derived>::Rows, ColsAtCompileTime = traits<Derived>::Cols,
           IsVectorAtCompileTime = (traits<Derived>::Rows == 1 || traits<Derived>::Cols == 1) };
    typedef Matrix<Scalar, traits<Derived>::Rows, traits<Derived>::Cols> PlainObject;
    typedef Matrix<Scalar, traits<Derived>::Cols, traits<Derived>::Rows> TransposeReturn;

    Derived& derived() { return *static_cast<Derived*>(this); }
    const Derived& derived() const { return *static_cast<const Derived*>(this); }
    int rows() const { return derived().rows_(); }
    int cols() const { return derived().cols_(); }
    int size() const { return rows() * cols(); }

    // coefficient access
    Scalar coeff(int i, int j) const { return derived().get_(i, j); }
    Scalar& coeffRef(int i, int j) { return derived().ref_(i, j); }
    Scalar operator()(int i, int j) const { return derived().get_(i, j); }
    Scalar& operator()(int i, int j) { return derived().ref_(i, j); }
    Scalar operator()(int i) const { return vget(i); }
    Scalar& operator()(int i) { return vref(i); }
    Scalar operator[](int i) const { return vget(i); }
    Scalar& operator[](int i) { return vref(i); }
    Scalar x() const { return vget(0); } Scalar y() const { return vget(1); } Scalar z() const { return vget(2); } Scalar w() const { return vget(3); }
    Scalar& x() { return vref(0); } Scalar& y() { return vref(1); } Scalar& z() { return vref(2); } Scalar& w() { return vref(3); }

    PlainObject eval() const { return PlainObject(*this); }

    // ---- assignment-like (element-wise through a temporary: aliasing-safe)
    template <class O> Derived& assign_from(const MatrixBase<O>& o) {
        const Matrix<Scalar, Dynamic, Dynamic> t(o);
        const bool fixed_vec_t = (traits<Derived>::Rows == 1 && t.cols() == 1 && t.rows() != 1) || (traits<Derived>::Cols == 1 && t.rows() == 1 && t.cols() != 1);
        if (fixed_vec_t) {                          // a row vector assigned to a column vector (or back): Eigen transposes vectors automatically
            derived().resize_like_(t.cols(), t.rows());
            for (int k = 0; k < t.size(); ++k) vref(k) = t.vget(k);
            return derived();
        }
        derived().resize_like_(t.rows(), t.cols());
        assert(rows() == t.rows() && cols() == t.cols());
        for (int j = 0; j < cols(); ++j) for (int i = 0; i < rows(); ++i) derived().ref_(i, j) = t.get_(i, j);
        return derived();
    }
    template <class O> Derived& operator+=(const MatrixBase<O>& o) {
        const Matrix<Scalar, Dynamic, Dynamic> t(o);
        assert(rows() == t.rows() && cols() == t.cols());
        for (int j = 0; j < cols(); ++j) for (int i = 0; i < rows(); ++i) derived().ref_(i, j) += t.get_(i, j);
        return derived();
    }
    template <class O> Derived& operator-=(const MatrixBase<O>& o) {
        const Matrix<Scalar, Dynamic, Dynamic> t(o);
        assert(rows() == t.rows() && cols() == t.cols());
        for (int j = 0; j < cols(); ++j) for (int i = 0; i < rows(); ++i) derived().ref_(i, j) -= t.get_(i, j);
        return derived();
    }
    Derived& operator*=(const Scalar& s) { for (int j = 0; j < cols(); ++j) for (int i = 0; i < rows(); ++i) derived().ref_(i, j) *= s; return derived(); }
    Derived& operator/=(const Scalar& s) { for (int j = 0; j < cols(); ++j) for (int i = 0; i < rows(); ++i) derived().ref_(i, j) /= s; return derived(); }
    Derived& setZero() { for (int j = 0; j < cols(); ++j) for (int i = 0; i < rows(); ++i) derived().ref_(i, j) = Scalar(0); return derived(); }
    Derived& setConstant(const Scalar& v) { for (int j = 0; j < cols(); ++j) for (int i = 0; i < rows(); ++i) derived().ref_(i, j) = v; return derived(); }
    Derived& setIdentity() { for (int j = 0; j < cols(); ++j) for (int i = 0; i < rows(); ++i) derived().ref_(i, j) = Scalar(i == j ? 1 : 0); return derived(); }

    // ---- reductions
    Scalar squaredNorm() const { Scalar s(0); for (int j = 0; j < cols(); ++j) for (int i = 0; i < rows(); ++i) { const Scalar v = coeff(i, j); s += v * v; } return s; }
    Scalar norm() const { using std::sqrt; return sqrt(squaredNorm()); }
    Scalar sum() const { Scalar s(0); for (int j = 0; j < cols(); ++j) for (int i = 0; i < rows(); ++i) s += coeff(i, j); return s; }
    Scalar trace() const { Scalar s(0); for (int i = 0; i < rows() && i < cols(); ++i) s += coeff(i, i); return s; }
    Scalar maxCoeff() const { Scalar m = coeff(0, 0); for (int j = 0; j < cols(); ++j) for (int i = 0; i < rows(); ++i) if (coeff(i, j) > m) m = coeff(i, j); return m; }
    Scalar minCoeff() const { Scalar m = coeff(0, 0); for (int j = 0; j < cols(); ++j) for (int i = 0; i < rows(); ++i) if (coeff(i, j) < m) m = coeff(i, j); return m; }
    PlainObject normalized() const { PlainObject r(*this); const Scalar n = norm(); if (n > Scalar(0)) r /= n; return r; }
    void normalize() { const Scalar n = norm(); if (n > Scalar(0)) (*this) /= n; }
    template <class O> Scalar dot(const MatrixBase<O>& o) const { assert(size() == o.size()); Scalar s(0); for (int k = 0; k < size(); ++k) s += vget(k) * o.vget(k); return s; }
    template <class O> Matrix<Scalar, 3, 1> cross(const MatrixBase<O>& o) const {
        Matrix<Scalar, 3, 1> r;
        const Scalar a0 = vget(0), a1 = vget(1), a2 = vget(2), b0 = o.vget(0), b1 = o.vget(1), b2 = o.vget(2);
        r(0) = a1 * b2 - a2 * b1; r(1) = a2 * b0 - a0 * b2; r(2) = a0 * b1 - a1 * b0;
        return r;
    }
    TransposeReturn transpose() const { TransposeReturn t; t.resize_like_(cols(), rows()); for (int j = 0; j < cols(); ++j) for (int i = 0; i < rows(); ++i) t.ref_(j, i) = coeff(i, j); return t; }
    template <class T> Matrix<T, traits<Derived>::Rows, traits<Derived>::Cols> cast() const {
        Matrix<T, traits<Derived>::Rows, traits<Derived>::Cols> r; r.resize_like_(rows(), cols());
        for (int j = 0; j < cols(); ++j) for (int i = 0; i < rows(); ++i) r.ref_(i, j) = T(coeff(i, j));
        return r;
    }
    PlainObject cwiseSqrt() const { using std::sqrt; PlainObject r(*this); for (int j = 0; j < cols(); ++j) for (int i = 0; i < rows(); ++i) r.ref_(i, j) = sqrt(coeff(i, j)); return r; }
    PlainObject cwiseAbs() const { using std::abs; PlainObject r(*this); for (int j = 0; j < cols(); ++j) for (int i = 0; i < rows(); ++i) r.ref_(i, j) = abs(coeff(i, j)); return r; }
    template <class O> PlainObject cwiseProduct(const MatrixBase<O>& o) const { PlainObject r(*this); for (int j = 0; j < cols(); ++j) for (int i = 0; i < rows(); ++i) r.ref_(i, j) = coeff(i, j) * o.coeff(i, j); return r; }
    ArrayWrap<Scalar> array() const { ArrayWrap<Scalar> a(size()); for (int k = 0; k < size(); ++k) a.v[k] = vget(k); return a; }
    DiagonalWrap<Scalar> asDiagonal() const { DiagonalWrap<Scalar> d; d.v.resize(size()); for (int k = 0; k < size(); ++k) d.v[k] = vget(k); return d; }
    Matrix<Scalar, Dynamic, 1> diagonal() const { Matrix<Scalar, Dynamic, 1> d(rows() < cols() ? rows() : cols()); for (int k = 0; k < d.size(); ++k) d(k) = coeff(k, k); return d; }
    PlainObject inverse() const;             // Gauss-Jordan with partial pivoting (square)
    Scalar determinant() const;
    bool allFinite() const { for (int j = 0; j < cols(); ++j) for (int i = 0; i < rows(); ++i) if (!std::isfinite((double)coeff(i, j))) return false; return true; }
    bool hasNaN() const { for (int j = 0; j < cols(); ++j) for (int i = 0; i < rows(); ++i) if (coeff(i, j) != coeff(i, j)) return true; return false; }

    // ---- blocks: views on non-const objects, plain copies on const ones
    template <int R, int C> Block<Derived, R, C> block(int i, int j) { return Block<Derived, R, C>(derived(), i, j, R, C); }
    template <int R, int C> Matrix<Scalar, R, C> block(int i, int j) const { return copy_block<R, C>(i, j, R, C); }
    Block<Derived, Dynamic, Dynamic> block(int i, int j, int r, int c) { return Block<Derived, Dynamic, Dynamic>(derived(), i, j, r, c); }
    Matrix<Scalar, Dynamic, Dynamic> block(int i, int j, int r, int c) const { return copy_block<Dynamic, Dynamic>(i, j, r, c); }
    template <int R, int C> Block<Derived, R, C> topLeftCorner() { return block<R, C>(0, 0); }
    template <int R, int C> Matrix<Scalar, R, C> topLeftCorner() const { return block<R, C>(0, 0); }
    template <int R, int C> Block<Derived, R, C> bottomRightCorner() { return block<R, C>(rows() - R, cols() - C); }
    template <int R, int C> Matrix<Scalar, R, C> bottomRightCorner() const { return block<R, C>(rows() - R, cols() - C); }
    template <int R, int C> Block<Derived, R, C> topRightCorner() { return block<R, C>(0, cols() - C); }
    template <int R, int C> Matrix<Scalar, R, C> topRightCorner() const { return block<R, C>(0, cols() - C); }
    template <int R, int C> Block<Derived, R, C> bottomLeftCorner() { return block<R, C>(rows() - R, 0); }
    template <int R, int C> Matrix<Scalar, R, C> bottomLeftCorner() const { return block<R, C>(rows() - R, 0); }
    Block<Derived, Dynamic, Dynamic> topLeftCorner(int r, int c) { return block(0, 0, r, c); }
    Matrix<Scalar, Dynamic, Dynamic> topLeftCorner(int r, int c) const { return block(0, 0, r, c); }
    Block<Derived, Dynamic, Dynamic> bottomRightCorner(int r, int c) { return block(rows() - r, cols() - c, r, c); }
    Matrix<Scalar, Dynamic, Dynamic> bottomRightCorner(int r, int c) const { return block(rows() - r, cols() - c, r, c); }
    Block<Derived, traits<Derived>::Rows, 1> col(int j) { return Block<Derived, traits<Derived>::Rows, 1>(derived(), 0, j, rows(), 1); }
    Matrix<Scalar, traits<Derived>::Rows, 1> col(int j) const { return copy_block<traits<Derived>::Rows, 1>(0, j, rows(), 1); }
    Block<Derived, 1, traits<Derived>::Cols> row(int i) { return Block<Derived, 1, traits<Derived>::Cols>(derived(), i, 0, 1, cols()); }
    Matrix<Scalar, 1, traits<Derived>::Cols> row(int i) const { return copy_block<1, traits<Derived>::Cols>(i, 0, 1, cols()); }
    Block<Derived, traits<Derived>::Rows, Dynamic> rightCols(int n) { return Block<Derived, traits<Derived>::Rows, Dynamic>(derived(), 0, cols() - n, rows(), n); }
    Matrix<Scalar, traits<Derived>::Rows, Dynamic> rightCols(int n) const { return copy_block<traits<Derived>::Rows, Dynamic>(0, cols() - n, rows(), n); }
    Block<Derived, traits<Derived>::Rows, Dynamic> leftCols(int n) { return Block<Derived, traits<Derived>::Rows, Dynamic>(derived(), 0, 0, rows(), n); }
    Matrix<Scalar, traits<Derived>::Rows, Dynamic> leftCols(int n) const { return copy_block<traits<Derived>::Rows, Dynamic>(0, 0, rows(), n); }
    Block<Derived, traits<Derived>::Rows, Dynamic> middleCols(int j, int n) { return Block<Derived, traits<Derived>::Rows, Dynamic>(derived(), 0, j, rows(), n); }
    Matrix<Scalar, traits<Derived>::Rows, Dynamic> middleCols(int j, int n) const { return copy_block<traits<Derived>::Rows, Dynamic>(0, j, rows(), n); }
    Block<Derived, Dynamic, traits<Derived>::Cols> topRows(int n) { return Block<Derived, Dynamic, traits<Derived>::Cols>(derived(), 0, 0, n, cols()); }
    Matrix<Scalar, Dynamic, traits<Derived>::Cols> topRows(int n) const { return copy_block<Dynamic, traits<Derived>::Cols>(0, 0, n, cols()); }
    Block<Derived, Dynamic, traits<Derived>::Cols> bottomRows(int n) { return Block<Derived, Dynamic, traits<Derived>::Cols>(derived(), rows() - n, 0, n, cols()); }
    Matrix<Scalar, Dynamic, traits<Derived>::Cols> bottomRows(int n) const { return copy_block<Dynamic, traits<Derived>::Cols>(rows() - n, 0, n, cols()); }
    // vector segments (column or row vectors)
    enum { SegR_dyn = (traits<Derived>::Cols == 1) ? Dynamic : 1, SegC_dyn = (traits<Derived>::Cols == 1) ? 1 : Dynamic };
    Block<Derived, SegR_dyn, SegC_dyn> segment(int i, int n) { return is_col() ? Block<Derived, SegR_dyn, SegC_dyn>(derived(), i, 0, n, 1) : Block<Derived, SegR_dyn, SegC_dyn>(derived(), 0, i, 1, n); }
    Matrix<Scalar, SegR_dyn, SegC_dyn> segment(int i, int n) const { return is_col() ? copy_block<SegR_dyn, SegC_dyn>(i, 0, n, 1) : copy_block<SegR_dyn, SegC_dyn>(0, i, 1, n); }
    template <int N> Block<Derived, (traits<Derived>::Cols == 1 ? N : 1), (traits<Derived>::Cols == 1 ? 1 : N)> segment(int i) {
        typedef Block<Derived, (traits<Derived>::Cols == 1 ? N : 1), (traits<Derived>::Cols == 1 ? 1 : N)> B;
        return is_col() ? B(derived(), i, 0, N, 1) : B(derived(), 0, i, 1, N);
    }
    template <int N> Matrix<Scalar, (traits<Derived>::Cols == 1 ? N : 1), (traits<Derived>::Cols == 1 ? 1 : N)> segment(int i) const {
        return is_col() ? copy_block<(traits<Derived>::Cols == 1 ? N : 1), (traits<Derived>::Cols == 1 ? 1 : N)>(i, 0, N, 1)
                        : copy_block<(traits<Derived>::Cols == 1 ? N : 1), (traits<Derived>::Cols == 1 ? 1 : N)>(0, i, 1, N);
    }
    Block<Derived, SegR_dyn, SegC_dyn> head(int n) { return segment(0, n); }
    Matrix<Scalar, SegR_dyn, SegC_dyn> head(int n) const { return segment(0, n); }
    Block<Derived, SegR_dyn, SegC_dyn> tail(int n) { return segment(size() - n, n); }
    Matrix<Scalar, SegR_dyn, SegC_dyn> tail(int n) const { return segment(size() - n, n); }
    template <int N> auto head() -> decltype(this->template segment<N>(0)) { return this->template segment<N>(0); }
    template <int N> auto head() const -> decltype(this->template segment<N>(0)) { return this->template segment<N>(0); }
    template <int N> auto tail() -> decltype(this->template segment<N>(0)) { return this->template segment<N>(size() - N); }
    template <int N> auto tail() const -> decltype(this->template segment<N>(0)) { return this->template segment<N>(size() - N); }

    // ---- comma initialisation
    template <class U> CommaInitializer<Derived> operator<<(const U& v) { return CommaInitializer<Derived>(derived(), v); }

    // ---- statics (fixed sizes, or sized)
    static PlainObject Zero() { PlainObject m; m.setZero(); return m; }
    static PlainObject Zero(int r, int c) { PlainObject m; m.resize_like_(r, c); m.setZero(); return m; }
    static PlainObject Zero(int n) { PlainObject m; if (traits<Derived>::Cols == 1) m.resize_like_(n, 1); else m.resize_like_(1, n); m.setZero(); return m; }
    static PlainObject Ones() { PlainObject m; m.setConstant(Scalar(1)); return m; }
    static PlainObject Constant(const Scalar& v) { PlainObject m; m.setConstant(v); return m; }
    static PlainObject Identity() { PlainObject m; m.setIdentity(); return m; }
    static PlainObject Identity(int r, int c) { PlainObject m; m.resize_like_(r, c); m.setIdentity(); return m; }
    static PlainObject UnitX() { PlainObject m; m.setZero(); m(0) = Scalar(1); return m; }
    static PlainObject UnitY() { PlainObject m; m.setZero(); m(1) = Scalar(1); return m; }
    static PlainObject UnitZ() { PlainObject m; m.setZero(); m(2) = Scalar(1); return m; }

    // vector-style linear access (public: used across instantiations)
    Scalar vget(int k) const { return (cols() == 1) ? coeff(k, 0) : (rows() == 1 ? coeff(0, k) : coeff(k % rows(), k / rows())); }
    Scalar& vref(int k) { return (cols() == 1) ? coeffRef(k, 0) : (rows() == 1 ? coeffRef(0, k) : coeffRef(k % rows(), k / rows())); }

  protected:
    bool is_col() const { return traits<Derived>::Cols == 1 || (traits<Derived>::Rows != 1 && cols() == 1); }
    template <int R, int C> Matrix<Scalar, R, C> copy_block(int i0, int j0, int r, int c) const {
        assert(i0 >= 0 && j0 >= 0 && i0 + r <= rows() && j0 + c <= cols());
        Matrix<Scalar, R, C> m; m.resize_like_(r, c);
        for (int j = 0; j < c; ++j) for (int i = 0; i < r; ++i) m.ref_(i, j) = coeff(i0 + i, j0 + j);
        return m;
    }
};

// ------------------------------------------------------------------------------------------------ storage
namespace internal {
template <class S, int R, int C, bool Fixed = (R != Dynamic && C != Dynamic)> struct Storage;
template <class S, int R, int C> struct Storage<S, R, C, true> {
    S d[R * C > 0 ? R * C : 1];
    Storage() { for (int k = 0; k < R * C; ++k) d[k] = S(0); }       // (Eigen leaves them uninitialised; zero is a legal instance of that)
    int rows() const { return R; } int cols() const { return C; }
    void resize(int r, int c) { assert(r == R && c == C); (void)r; (void)c; }
    S* data() { return d; } const S* data() const { return d; }
};
template <class S, int R, int C> struct Storage<S, R, C, false> {
    std::vector<S> d; int r_, c_;
    Storage() : r_(R == Dynamic ? 0 : R), c_(C == Dynamic ? 0 : C) {}
    int rows() const { return r_; } int cols() const { return c_; }
    void resize(int r, int c) { assert((R == Dynamic || r == R) && (C == Dynamic || c == C)); if (r != r_ || c != c_) { r_ = r; c_ = c; d.assign((size_t)r * c, S(0)); } }
    S* data() { return d.data(); } const S* data() const { return d.data(); }
};
}  // namespace internal

// ------------------------------------------------------------------------------------------------ Matrix
template <class S, int R, int C, int Opt, int MR, int MC> struct traits<Matrix<S, R, C, Opt, MR, MC> > { typedef S Scalar; enum { Rows = R, Cols = C, Options = Opt }; };

template <class S, int R, int C, int Opt, int MR, int MC>
class Matrix : public MatrixBase<Matrix<S, R, C, Opt, MR, MC> > {
  public:
    typedef S Scalar;
    typedef MatrixBase<Matrix> Base;
    enum { IsRowMajor = (Opt & RowMajor) ? 1 : 0 };

    Matrix() {}
    Matrix(const Matrix& o) : st(o.st) {}
    template <class O> Matrix(const MatrixBase<O>& o) {
        if ((R == 1 && o.cols() == 1 && o.rows() != 1) || (C == 1 && o.rows() == 1 && o.cols() != 1)) {      // vector from a transposed vector
            st.resize(o.cols(), o.rows());
            for (int k = 0; k < o.size(); ++k) this->vref(k) = o.vget(k);
            return;
        }
        st.resize(o.rows(), o.cols());
        for (int j = 0; j < o.cols(); ++j) for (int i = 0; i < o.rows(); ++i) ref_(i, j) = o.coeff(i, j);
    }
    Matrix(const ArrayWrap<S>& a) { resize_like_(R == 1 ? 1 : (int)a.v.size(), R == 1 ? (int)a.v.size() : 1); for (int k = 0; k < (int)a.v.size(); ++k) this->vref(k) = a.v[k]; }
    // size constructors (dynamic) -- for fixed sizes a single integer is the size too (a no-op), as in Eigen
    template <class I, typename std::enable_if<std::is_integral<I>::value, int>::type = 0>
    explicit Matrix(I n) { if (R == Dynamic && C == Dynamic) st.resize((int)n, 1); else if (R == Dynamic) st.resize((int)n, C); else if (C == Dynamic) st.resize(R, (int)n); }
    // two arguments: (rows, cols) for dynamic sizes, the two coefficients for a fixed 2-vector
    template <class A, class B, typename std::enable_if<std::is_convertible<A, S>::value && std::is_convertible<B, S>::value, int>::type = 0>
    Matrix(const A& a, const B& b) { init2(a, b, std::integral_constant<bool, (R != Dynamic && C != Dynamic)>()); }
    // three / four coefficients (fixed vectors)
    template <class A, class B, class D, typename std::enable_if<std::is_convertible<A, S>::value && std::is_convertible<B, S>::value && std::is_convertible<D, S>::value, int>::type = 0>
    Matrix(const A& a, const B& b, const D& c) { static_assert(R * C == 3, "3-coefficient constructor on a non-3-vector"); st.d[0] = S(a); st.d[1] = S(b); st.d[2] = S(c); }
    template <class A, class B, class D, class E,
              typename std::enable_if<std::is_convertible<A, S>::value && std::is_convertible<B, S>::value && std::is_convertible<D, S>::value && std::is_convertible<E, S>::value, int>::type = 0>
    Matrix(const A& a, const B& b, const D& c, const E& d) { static_assert(R * C == 4, "4-coefficient constructor on a non-4-vector"); st.d[0] = S(a); st.d[1] = S(b); st.d[2] = S(c); st.d[3] = S(d); }

    Matrix& operator=(const Matrix& o) { st = o.st; return *this; }
    template <class O> Matrix& operator=(const MatrixBase<O>& o) { return this->assign_from(o); }
    Matrix& operator=(const ArrayWrap<S>& a) { *this = Matrix(a); return *this; }

    void resize(int r, int c) { st.resize(r, c); }
    void resize(int n) { if (C == 1 || (R == Dynamic && C == Dynamic)) st.resize(n, C == 1 ? 1 : 1); else st.resize(1, n); }
    void conservativeResize(int r, int c) { Matrix t; t.st.resize(r, c); for (int j = 0; j < c && j < cols_(); ++j) for (int i = 0; i < r && i < rows_(); ++i) t.ref_(i, j) = get_(i, j); *this = t; }
    S* data() { return st.data(); }
    const S* data() const { return st.data(); }

    // CRTP hooks
    int rows_() const { return st.rows(); }
    int cols_() const { return st.cols(); }
    S get_(int i, int j) const { assert(i >= 0 && j >= 0 && i < rows_() && j < cols_()); return st.data()[IsRowMajor ? (size_t)i * cols_() + j : (size_t)j * rows_() + i]; }
    S& ref_(int i, int j) { assert(i >= 0 && j >= 0 && i < rows_() && j < cols_()); return st.data()[IsRowMajor ? (size_t)i * cols_() + j : (size_t)j * rows_() + i]; }
    void resize_like_(int r, int c) { st.resize(r, c); }

  private:
    template <class A, class B> void init2(const A& a, const B& b, std::true_type) { static_assert(R * C == 2 || (R != Dynamic && C != Dynamic), ""); if (R * C == 2) { st.d[0] = S(a); st.d[1] = S(b); } }
    template <class A, class B> void init2(const A& a, const B& b, std::false_type) { st.resize((int)a, (int)b); }
    internal::Storage<S, R, C> st;
};

typedef Matrix<double, 2, 1> Vector2d; typedef Matrix<double, 3, 1> Vector3d; typedef Matrix<double, 4, 1> Vector4d;
typedef Matrix<float, 3, 1> Vector3f; typedef Matrix<float, 4, 1> Vector4f;
typedef Matrix<double, 2, 2> Matrix2d; typedef Matrix<double, 3, 3> Matrix3d; typedef Matrix<double, 4, 4> Matrix4d;
typedef Matrix<float, 3, 3> Matrix3f; typedef Matrix<float, 4, 4> Matrix4f;
typedef Matrix<double, Dynamic, Dynamic> MatrixXd; typedef Matrix<double, Dynamic, 1> VectorXd; typedef Matrix<double, 1, Dynamic> RowVectorXd;
typedef Matrix<float, Dynamic, Dynamic> MatrixXf; typedef Matrix<float, Dynamic, 1> VectorXf;
typedef Matrix<int, Dynamic, 1> VectorXi; typedef Matrix<int, Dynamic, Dynamic> MatrixXi;

// ------------------------------------------------------------------------------------------------ Block (a view)
template <class Xpr, int R, int C> struct traits<Block<Xpr, R, C> > { typedef typename traits<Xpr>::Scalar Scalar; enum { Rows = R, Cols = C, Options = 0 }; };
template <class Xpr, int R, int C>
class Block : public MatrixBase<Block<Xpr, R, C> > {
  public:
    typedef typename traits<Xpr>::Scalar Scalar;
    Block(Xpr& x, int i0, int j0, int r, int c) : x_(&x), i0_(i0), j0_(j0), r_(r), c_(c) { assert(i0 >= 0 && j0 >= 0 && i0 + r <= x.rows() && j0 + c <= x.cols()); }
    Block(const Block& o) : x_(o.x_), i0_(o.i0_), j0_(o.j0_), r_(o.r_), c_(o.c_) {}
    Block& operator=(const Block& o) { return this->assign_from(o); }          // element-wise, never a re-binding
    template <class O> Block& operator=(const MatrixBase<O>& o) { return this->assign_from(o); }
    int rows_() const { return r_; } int cols_() const { return c_; }
    Scalar get_(int i, int j) const { return static_cast<const Xpr*>(x_)->coeff(i0_ + i, j0_ + j); }
    Scalar& ref_(int i, int j) { return x_->coeffRef(i0_ + i, j0_ + j); }
    void resize_like_(int r, int c) { assert(r == r_ && c == c_); (void)r; (void)c; }
  private:
    Xpr* x_; int i0_, j0_, r_, c_;
};

// ------------------------------------------------------------------------------------------------ Map
template <class S, int R, int C, int Opt, int MR, int MC, int MapOpt, class Stride>
struct traits<Map<Matrix<S, R, C, Opt, MR, MC>, MapOpt, Stride> > { typedef S Scalar; enum { Rows = R, Cols = C, Options = Opt }; };
template <class S, int R, int C, int Opt, int MR, int MC, int MapOpt, class Stride>
struct traits<Map<const Matrix<S, R, C, Opt, MR, MC>, MapOpt, Stride> > { typedef S Scalar; enum { Rows = R, Cols = C, Options = Opt }; };

template <class S, int R, int C, int Opt, int MR, int MC, int MapOpt, class Stride>
class Map<Matrix<S, R, C, Opt, MR, MC>, MapOpt, Stride> : public MatrixBase<Map<Matrix<S, R, C, Opt, MR, MC>, MapOpt, Stride> > {
  public:
    typedef S Scalar;
    enum { IsRowMajor = (Opt & RowMajor) ? 1 : 0 };
    explicit Map(S* p) : p_(p), r_(R), c_(C) { static_assert(R != Dynamic && C != Dynamic, "sizes needed"); }
    Map(S* p, int n) : p_(p), r_(C == 1 || R != 1 ? n : 1), c_(C == 1 || R != 1 ? 1 : n) { if (R != Dynamic && C != Dynamic) { r_ = R; c_ = C; } }
    Map(S* p, int r, int c) : p_(p), r_(r), c_(c) {}
    Map(const Map& o) : p_(o.p_), r_(o.r_), c_(o.c_) {}
    Map& operator=(const Map& o) { return this->assign_from(o); }
    template <class O> Map& operator=(const MatrixBase<O>& o) { return this->assign_from(o); }
    int rows_() const { return r_; } int cols_() const { return c_; }
    S get_(int i, int j) const { return p_[IsRowMajor ? (size_t)i * c_ + j : (size_t)j * r_ + i]; }
    S& ref_(int i, int j) { return p_[IsRowMajor ? (size_t)i * c_ + j : (size_t)j * r_ + i]; }
    void resize_like_(int r, int c) { assert(r == r_ && c == c_); (void)r; (void)c; }
    S* data() { return p_; } const S* data() const { return p_; }
  private:
    S* p_; int r_, c_;
};
template <class S, int R, int C, int Opt, int MR, int MC, int MapOpt, class Stride>
class Map<const Matrix<S, R, C, Opt, MR, MC>, MapOpt, Stride> : public MatrixBase<Map<const Matrix<S, R, C, Opt, MR, MC>, MapOpt, Stride> > {
  public:
    typedef S Scalar;
    enum { IsRowMajor = (Opt & RowMajor) ? 1 : 0 };
    explicit Map(const S* p) : p_(p), r_(R), c_(C) { static_assert(R != Dynamic && C != Dynamic, "sizes needed"); }
    Map(const S* p, int n) : p_(p), r_(C == 1 || R != 1 ? n : 1), c_(C == 1 || R != 1 ? 1 : n) { if (R != Dynamic && C != Dynamic) { r_ = R; c_ = C; } }
    Map(const S* p, int r, int c) : p_(p), r_(r), c_(c) {}
    int rows_() const { return r_; } int cols_() const { return c_; }
    S get_(int i, int j) const { return p_[IsRowMajor ? (size_t)i * c_ + j : (size_t)j * r_ + i]; }
    S& ref_(int, int) { static S dummy; assert(!"write through a Map<const>"); return dummy; }
    void resize_like_(int r, int c) { assert(r == r_ && c == c_); (void)r; (void)c; }
    const S* data() const { return p_; }
  private:
    const S* p_; int r_, c_;
};

// ------------------------------------------------------------------------------------------------ CommaInitializer
template <class Derived> class CommaInitializer {
  public:
    typedef typename traits<Derived>::Scalar Scalar;
    template <class U> CommaInitializer(Derived& m, const U& v) : m_(m), row_(0), col_(0), brows_(1) { put(v, typename std::is_convertible<U, Scalar>::type()); }
    template <class U> CommaInitializer& operator,(const U& v) { put(v, typename std::is_convertible<U, Scalar>::type()); return *this; }
    Derived& finished() { return m_; }
  private:
    template <class U> void put(const U& v, std::true_type) {
        if (col_ == m_.cols()) { row_ += brows_; col_ = 0; brows_ = 1; }
        assert(row_ < m_.rows() && col_ < m_.cols());
        m_.coeffRef(row_, col_++) = Scalar(v);
    }
    template <class O> void put(const MatrixBase<O>& b, std::false_type) {
        if (col_ == m_.cols()) { row_ += brows_; col_ = 0; brows_ = b.rows(); }
        if (col_ == 0) brows_ = b.rows();
        assert(row_ + b.rows() <= m_.rows() && col_ + b.cols() <= m_.cols());
        for (int j = 0; j < b.cols(); ++j) for (int i = 0; i < b.rows(); ++i) m_.coeffRef(row_ + i, col_ + j) = b.coeff(i, j);
        col_ += b.cols();
    }
    Derived& m_; int row_, col_, brows_;
};

// ------------------------------------------------------------------------------------------------ arithmetic (eager)
#define GLIO_ME_RES(A, B) Matrix<typename traits<A>::Scalar, internal::pick_dim<traits<A>::Rows, traits<B>::Rows>::value, internal::pick_dim<traits<A>::Cols, traits<B>::Cols>::value>
template <class A, class B> GLIO_ME_RES(A, B) operator+(const MatrixBase<A>& a, const MatrixBase<B>& b) {
    assert(a.rows() == b.rows() && a.cols() == b.cols());
    GLIO_ME_RES(A, B) r; r.resize_like_(a.rows(), a.cols());
    for (int j = 0; j < a.cols(); ++j) for (int i = 0; i < a.rows(); ++i) r.ref_(i, j) = a.coeff(i, j) + b.coeff(i, j);
    return r;
}
template <class A, class B> GLIO_ME_RES(A, B) operator-(const MatrixBase<A>& a, const MatrixBase<B>& b) {
    assert(a.rows() == b.rows() && a.cols() == b.cols());
    GLIO_ME_RES(A, B) r; r.resize_like_(a.rows(), a.cols());
    for (int j = 0; j < a.cols(); ++j) for (int i = 0; i < a.rows(); ++i) r.ref_(i, j) = a.coeff(i, j) - b.coeff(i, j);
    return r;
}
#undef GLIO_ME_RES
template <class A> typename MatrixBase<A>::PlainObject operator-(const MatrixBase<A>& a) {
    typename MatrixBase<A>::PlainObject r; r.resize_like_(a.rows(), a.cols());
    for (int j = 0; j < a.cols(); ++j) for (int i = 0; i < a.rows(); ++i) r.ref_(i, j) = -a.coeff(i, j);
    return r;
}
template <class A, class B> Matrix<typename traits<A>::Scalar, traits<A>::Rows, traits<B>::Cols> operator*(const MatrixBase<A>& a, const MatrixBase<B>& b) {
    typedef typename traits<A>::Scalar S;
    assert(a.cols() == b.rows());
    Matrix<S, traits<A>::Rows, traits<B>::Cols> r; r.resize_like_(a.rows(), b.cols());
    for (int j = 0; j < b.cols(); ++j) for (int i = 0; i < a.rows(); ++i) {
        S s(0);
        for (int k = 0; k < a.cols(); ++k) s += a.coeff(i, k) * b.coeff(k, j);
        r.ref_(i, j) = s;
    }
    return r;
}
template <class A> typename MatrixBase<A>::PlainObject operator*(const MatrixBase<A>& a, const typename traits<A>::Scalar& s) {
    typename MatrixBase<A>::PlainObject r; r.resize_like_(a.rows(), a.cols());
    for (int j = 0; j < a.cols(); ++j) for (int i = 0; i < a.rows(); ++i) r.ref_(i, j) = a.coeff(i, j) * s;
    return r;
}
template <class A> typename MatrixBase<A>::PlainObject operator*(const typename traits<A>::Scalar& s, const MatrixBase<A>& a) {
    typename MatrixBase<A>::PlainObject r; r.resize_like_(a.rows(), a.cols());
    for (int j = 0; j < a.cols(); ++j) for (int i = 0; i < a.rows(); ++i) r.ref_(i, j) = s * a.coeff(i, j);
    return r;
}
template <class A> typename MatrixBase<A>::PlainObject operator/(const MatrixBase<A>& a, const typename traits<A>::Scalar& s) {
    typename MatrixBase<A>::PlainObject r; r.resize_like_(a.rows(), a.cols());
    for (int j = 0; j < a.cols(); ++j) for (int i = 0; i < a.rows(); ++i) r.ref_(i, j) = a.coeff(i, j) / s;
    return r;
}
template <class A> std::ostream& operator<<(std::ostream& os, const MatrixBase<A>& a) {
    for (int i = 0; i < a.rows(); ++i) { for (int j = 0; j < a.cols(); ++j) os << (j ? " " : "") << a.coeff(i, j); if (i + 1 < a.rows()) os << "\n"; }
    return os;
}

// ------------------------------------------------------------------------------------------------ inverse / determinant
template <class Derived> typename MatrixBase<Derived>::PlainObject MatrixBase<Derived>::inverse() const {
    using std::abs;
    const int n = rows(); assert(n == cols());
    Matrix<Scalar, Dynamic, Dynamic> a(*this), inv = Matrix<Scalar, Dynamic, Dynamic>::Identity(n, n);
    for (int c = 0; c < n; ++c) {
        int p = c;
        for (int i = c + 1; i < n; ++i) if (abs(a(i, c)) > abs(a(p, c))) p = i;
        if (p != c) for (int j = 0; j < n; ++j) { Scalar t = a(c, j); a(c, j) = a(p, j); a(p, j) = t; t = inv(c, j); inv(c, j) = inv(p, j); inv(p, j) = t; }
        const Scalar d = Scalar(1) / a(c, c);
        for (int j = 0; j < n; ++j) { a(c, j) *= d; inv(c, j) *= d; }
        for (int i = 0; i < n; ++i) if (i != c) {
            const Scalar f = a(i, c);
            if (f == Scalar(0)) continue;
            for (int j = 0; j < n; ++j) { a(i, j) -= f * a(c, j); inv(i, j) -= f * inv(c, j); }
        }
    }
    return PlainObject(inv);
}
template <class Derived> typename MatrixBase<Derived>::Scalar MatrixBase<Derived>::determinant() const {
    using std::abs;
    const int n = rows(); assert(n == cols());
    Matrix<Scalar, Dynamic, Dynamic> a(*this);
    Scalar det(1);
    for (int c = 0; c < n; ++c) {
        int p = c;
        for (int i = c + 1; i < n; ++i) if (abs(a(i, c)) > abs(a(p, c))) p = i;
        if (a(p, c) == Scalar(0)) return Scalar(0);
        if (p != c) { for (int j = 0; j < n; ++j) { Scalar t = a(c, j); a(c, j) = a(p, j); a(p, j) = t; } det = -det; }
        det *= a(c, c);
        for (int i = c + 1; i < n; ++i) { const Scalar f = a(i, c) / a(c, c); for (int j = c; j < n; ++j) a(i, j) -= f * a(c, j); }
    }
    return det;
}

// ------------------------------------------------------------------------------------------------ Array / Diagonal wrappers
struct BoolArray { std::vector<char> b;
    template <class S> ArrayWrap<S> select(const ArrayWrap<S>& then_, const S& else_) const { ArrayWrap<S> r((int)b.size()); for (size_t k = 0; k < b.size(); ++k) r.v[k] = b[k] ? then_.v[k] : else_; return r; }
    template <class S> ArrayWrap<S> select(const ArrayWrap<S>& then_, int else_) const { return select(then_, S(else_)); }
    template <class S> ArrayWrap<S> select(const ArrayWrap<S>& then_, const ArrayWrap<S>& else_) const { ArrayWrap<S> r((int)b.size()); for (size_t k = 0; k < b.size(); ++k) r.v[k] = b[k] ? then_.v[k] : else_.v[k]; return r; }
};
template <class S> class ArrayWrap {
  public:
    std::vector<S> v;
    ArrayWrap() {}
    explicit ArrayWrap(int n) : v((size_t)n, S(0)) {}
    BoolArray operator>(const S& t) const { BoolArray r; r.b.resize(v.size()); for (size_t k = 0; k < v.size(); ++k) r.b[k] = v[k] > t; return r; }
    BoolArray operator<(const S& t) const { BoolArray r; r.b.resize(v.size()); for (size_t k = 0; k < v.size(); ++k) r.b[k] = v[k] < t; return r; }
    ArrayWrap inverse() const { ArrayWrap r((int)v.size()); for (size_t k = 0; k < v.size(); ++k) r.v[k] = S(1) / v[k]; return r; }
    ArrayWrap sqrt() const { using std::sqrt; ArrayWrap r((int)v.size()); for (size_t k = 0; k < v.size(); ++k) r.v[k] = sqrt(v[k]); return r; }
    ArrayWrap abs() const { using std::abs; ArrayWrap r((int)v.size()); for (size_t k = 0; k < v.size(); ++k) r.v[k] = abs(v[k]); return r; }
    Matrix<S, Dynamic, 1> matrix() const { return Matrix<S, Dynamic, 1>(*this); }
};
template <class S> class DiagonalWrap { public: std::vector<S> v; };
template <class S, class B> Matrix<S, Dynamic, traits<B>::Cols> operator*(const DiagonalWrap<S>& d, const MatrixBase<B>& b) {
    assert((int)d.v.size() == b.rows());
    Matrix<S, Dynamic, traits<B>::Cols> r; r.resize_like_(b.rows(), b.cols());
    for (int j = 0; j < b.cols(); ++j) for (int i = 0; i < b.rows(); ++i) r.ref_(i, j) = d.v[i] * b.coeff(i, j);
    return r;
}
template <class A, class S> Matrix<S, traits<A>::Rows, Dynamic> operator*(const MatrixBase<A>& a, const DiagonalWrap<S>& d) {
    assert((int)d.v.size() == a.cols());
    Matrix<S, traits<A>::Rows, Dynamic> r; r.resize_like_(a.rows(), a.cols());
    for (int j = 0; j < a.cols(); ++j) for (int i = 0; i < a.rows(); ++i) r.ref_(i, j) = a.coeff(i, j) * d.v[j];
    return r;
}

// ------------------------------------------------------------------------------------------------ LLT
template <class MatrixType, int UpLo = Lower> class LLT {
  public:
    typedef typename traits<MatrixType>::Scalar Scalar;
    LLT() : ok_(false) {}
    template <class O> explicit LLT(const MatrixBase<O>& a) { compute(a); }
    template <class O> LLT& compute(const MatrixBase<O>& a) {
        using std::sqrt;
        const int n = a.rows();
        L_ = MatrixType(a); ok_ = true;
        for (int j = 0; j < n; ++j) {
            Scalar d = L_(j, j);
            for (int k = 0; k < j; ++k) d -= L_(j, k) * L_(j, k);
            if (!(d > Scalar(0))) ok_ = false;
            const Scalar ljj = sqrt(d);
            L_(j, j) = ljj;
            for (int i = j + 1; i < n; ++i) {
                Scalar s = L_(i, j);
                for (int k = 0; k < j; ++k) s -= L_(i, k) * L_(j, k);
                L_(i, j) = s / ljj;
            }
            for (int i = 0; i < j; ++i) L_(i, j) = Scalar(0);
        }
        return *this;
    }
    MatrixType matrixL() const { return L_; }
    typename MatrixBase<MatrixType>::TransposeReturn matrixU() const { return L_.transpose(); }
    int info() const { return ok_ ? 0 : 1; }
    template <class B> Matrix<Scalar, traits<MatrixType>::Rows, traits<B>::Cols> solve(const MatrixBase<B>& b) const {
        const int n = L_.rows();
        Matrix<Scalar, traits<MatrixType>::Rows, traits<B>::Cols> x(b);
        for (int c = 0; c < x.cols(); ++c) {
            for (int i = 0; i < n; ++i) { Scalar s = x(i, c); for (int k = 0; k < i; ++k) s -= L_(i, k) * x(k, c); x(i, c) = s / L_(i, i); }
            for (int i = n - 1; i >= 0; --i) { Scalar s = x(i, c); for (int k = i + 1; k < n; ++k) s -= L_(k, i) * x(k, c); x(i, c) = s / L_(i, i); }
        }
        return x;
    }
  private:
    MatrixType L_; bool ok_;
};
enum ComputationInfo { Success = 0, NumericalIssue = 1, NoConvergence = 2, InvalidInput = 3 };

// ------------------------------------------------------------------------------------------------ SelfAdjointEigenSolver
// Cyclic Jacobi rotations on the symmetric matrix (lower triangle mirrored), eigenvalues ascending, eigenvectors in
// the columns -- the contract of Eigen's SelfAdjointEigenSolver (which uses tridiagonalisation + implicit QR).
template <class MatrixType> class SelfAdjointEigenSolver {
  public:
    typedef typename traits<MatrixType>::Scalar Scalar;
    typedef Matrix<Scalar, traits<MatrixType>::Rows, 1> RealVectorType;
    SelfAdjointEigenSolver() {}
    template <class O> explicit SelfAdjointEigenSolver(const MatrixBase<O>& a) { compute(a); }
    template <class O> SelfAdjointEigenSolver& compute(const MatrixBase<O>& a_) {
        const int n = a_.rows();
        std::vector<double> a((size_t)n * n), v((size_t)n * n, 0.0);
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) a[(size_t)i * n + j] = (double)(i >= j ? a_.coeff(i, j) : a_.coeff(j, i));   // lower triangle is read
        for (int i = 0; i < n; ++i) v[(size_t)i * n + i] = 1.0;
        for (int sweep = 0; sweep < 60; ++sweep) {
            double off = 0.0, diag = 0.0;
            for (int i = 0; i < n; ++i) { diag += a[(size_t)i * n + i] * a[(size_t)i * n + i]; for (int j = 0; j < i; ++j) off += a[(size_t)i * n + j] * a[(size_t)i * n + j]; }
            if (off <= 1e-36 * diag || off == 0.0) break;
            for (int p = 0; p < n - 1; ++p) for (int q = p + 1; q < n; ++q) {
                const double apq = a[(size_t)p * n + q];
                if (apq == 0.0) continue;
                const double app = a[(size_t)p * n + p], aqq = a[(size_t)q * n + q];
                const double theta = (aqq - app) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; ++k) {
                    const double akp = a[(size_t)k * n + p], akq = a[(size_t)k * n + q];
                    a[(size_t)k * n + p] = c * akp - s * akq; a[(size_t)k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {
                    const double apk = a[(size_t)p * n + k], aqk = a[(size_t)q * n + k];
                    a[(size_t)p * n + k] = c * apk - s * aqk; a[(size_t)q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    const double vkp = v[(size_t)k * n + p], vkq = v[(size_t)k * n + q];
                    v[(size_t)k * n + p] = c * vkp - s * vkq; v[(size_t)k * n + q] = s * vkp + c * vkq;
                }
            }
        }
        std::vector<int> ord(n);
        for (int i = 0; i < n; ++i) ord[i] = i;
        for (int i = 1; i < n; ++i) { const int o = ord[i]; int j = i - 1; while (j >= 0 && a[(size_t)ord[j] * n + ord[j]] > a[(size_t)o * n + o]) { ord[j + 1] = ord[j]; --j; } ord[j + 1] = o; }
        vals_.resize_like_(n, 1); vecs_.resize_like_(n, n);
        for (int k = 0; k < n; ++k) { vals_(k) = Scalar(a[(size_t)ord[k] * n + ord[k]]); for (int i = 0; i < n; ++i) vecs_(i, k) = Scalar(v[(size_t)i * n + ord[k]]); }
        return *this;
    }
    const RealVectorType& eigenvalues() const { return vals_; }
    const MatrixType& eigenvectors() const { return vecs_; }
    ComputationInfo info() const { return Success; }
  private:
    RealVectorType vals_; MatrixType vecs_;
};

// ------------------------------------------------------------------------------------------------ Quaternion
template <class D> struct quat_traits;
template <class S> struct quat_traits<Quaternion<S> > { typedef S Scalar; };
template <class Derived> class QuaternionBase {
  public:
    typedef typename quat_traits<Derived>::Scalar Scalar;
    Derived& derived() { return *static_cast<Derived*>(this); }
    const Derived& derived() const { return *static_cast<const Derived*>(this); }
    const Scalar& w() const { return derived().w_c(); } const Scalar& x() const { return derived().x_c(); }
    const Scalar& y() const { return derived().y_c(); } const Scalar& z() const { return derived().z_c(); }
    Matrix<Scalar, 3, 1> vec() const { return derived().vec_c(); }
    Quaternion<Scalar> inverse() const { return derived().inverse_c(); }
    Quaternion<Scalar> conjugate() const { return derived().conjugate_c(); }
    Quaternion<Scalar> normalized() const { return derived().normalized_c(); }
    Matrix<Scalar, 3, 3> toRotationMatrix() const { return derived().rot_c(); }
};
template <class S> class Quaternion : public QuaternionBase<Quaternion<S> > {
  public:
    typedef S Scalar;
    typedef Matrix<S, 3, 1> Vector3; typedef Matrix<S, 3, 3> Matrix3;
    Quaternion() : w_(S(0)), x_(S(0)), y_(S(0)), z_(S(0)) {}
    Quaternion(const S& w, const S& x, const S& y, const S& z) : w_(w), x_(x), y_(y), z_(z) {}
    Quaternion(const Quaternion& o) : w_(o.w_), x_(o.x_), y_(o.y_), z_(o.z_) {}
    Quaternion(const QuaternionBase<Quaternion>& o) : w_(o.derived().w_), x_(o.derived().x_), y_(o.derived().y_), z_(o.derived().z_) {}
    // hooks of QuaternionBase's forwards
    const S& w_c() const { return w_; } const S& x_c() const { return x_; } const S& y_c() const { return y_; } const S& z_c() const { return z_; }
    Matrix<S, 3, 1> vec_c() const { return vec(); } Quaternion inverse_c() const { return inverse(); } Quaternion conjugate_c() const { return conjugate(); }
    Quaternion normalized_c() const { return normalized(); } Matrix<S, 3, 3> rot_c() const { return toRotationMatrix(); }
    template <class O> explicit Quaternion(const MatrixBase<O>& m) { if (m.rows() == 3 && m.cols() == 3) fromRotationMatrix(m); else { x_ = m.vget(0); y_ = m.vget(1); z_ = m.vget(2); w_ = m.vget(3); } }
    explicit Quaternion(const AngleAxis<S>& aa) { *this = aa; }
    Quaternion& operator=(const Quaternion& o) { w_ = o.w_; x_ = o.x_; y_ = o.y_; z_ = o.z_; return *this; }
    template <class O> Quaternion& operator=(const MatrixBase<O>& m) { fromRotationMatrix(m); return *this; }
    Quaternion& operator=(const AngleAxis<S>& aa) {
        using std::sin; using std::cos;
        const S h = aa.angle() * S(0.5), s = sin(h);
        w_ = cos(h); x_ = s * aa.axis()(0); y_ = s * aa.axis()(1); z_ = s * aa.axis()(2);
        return *this;
    }
    const S& w() const { return w_; } const S& x() const { return x_; } const S& y() const { return y_; } const S& z() const { return z_; }
    S& w() { return w_; } S& x() { return x_; } S& y() { return y_; } S& z() { return z_; }
    Vector3 vec() const { return Vector3(x_, y_, z_); }
    Matrix<S, 4, 1> coeffs() const { return Matrix<S, 4, 1>(x_, y_, z_, w_); }
    S squaredNorm() const { return w_ * w_ + x_ * x_ + y_ * y_ + z_ * z_; }
    S norm() const { using std::sqrt; return sqrt(squaredNorm()); }
    void normalize() { const S n = norm(); w_ /= n; x_ /= n; y_ /= n; z_ /= n; }
    Quaternion normalized() const { Quaternion q(*this); q.normalize(); return q; }
    Quaternion conjugate() const { return Quaternion(w_, -x_, -y_, -z_); }
    Quaternion inverse() const {
        const S n2 = squaredNorm();
        if (n2 > S(0)) return Quaternion(w_ / n2, -x_ / n2, -y_ / n2, -z_ / n2);
        return Quaternion(S(0), S(0), S(0), S(0));
    }
    S dot(const Quaternion& o) const { return w_ * o.w_ + x_ * o.x_ + y_ * o.y_ + z_ * o.z_; }
    Quaternion operator*(const Quaternion& b) const {
        return Quaternion(w_ * b.w_ - x_ * b.x_ - y_ * b.y_ - z_ * b.z_,
                          w_ * b.x_ + x_ * b.w_ + y_ * b.z_ - z_ * b.y_,
                          w_ * b.y_ + y_ * b.w_ + z_ * b.x_ - x_ * b.z_,
                          w_ * b.z_ + z_ * b.w_ + x_ * b.y_ - y_ * b.x_);
    }
    Quaternion& operator*=(const Quaternion& b) { *this = *this * b; return *this; }
    // rotation of a vector: v + 2 w (q x v) + 2 q x (q x v)   (Eigen's _transformVector)
    template <class O> Vector3 operator*(const MatrixBase<O>& v_) const {
        const Vector3 v(v_.vget(0), v_.vget(1), v_.vget(2)), q(x_, y_, z_);
        Vector3 uv = q.cross(v);
        uv += uv;
        return v + w_ * uv + q.cross(uv);
    }
    Matrix3 toRotationMatrix() const {
        Matrix3 R;
        const S tx = S(2) * x_, ty = S(2) * y_, tz = S(2) * z_;
        const S twx = tx * w_, twy = ty * w_, twz = tz * w_, txx = tx * x_, txy = ty * x_, txz = tz * x_, tyy = ty * y_, tyz = tz * y_, tzz = tz * z_;
        R(0, 0) = S(1) - (tyy + tzz); R(0, 1) = txy - twz; R(0, 2) = txz + twy;
        R(1, 0) = txy + twz; R(1, 1) = S(1) - (txx + tzz); R(1, 2) = tyz - twx;
        R(2, 0) = txz - twy; R(2, 1) = tyz + twx; R(2, 2) = S(1) - (txx + tyy);
        return R;
    }
    Matrix3 matrix() const { return toRotationMatrix(); }
    template <class O> void fromRotationMatrix(const MatrixBase<O>& m) {
        using std::sqrt;
        S t = m.coeff(0, 0) + m.coeff(1, 1) + m.coeff(2, 2);
        if (t > S(0)) {
            t = sqrt(t + S(1)); w_ = S(0.5) * t; t = S(0.5) / t;
            x_ = (m.coeff(2, 1) - m.coeff(1, 2)) * t; y_ = (m.coeff(0, 2) - m.coeff(2, 0)) * t; z_ = (m.coeff(1, 0) - m.coeff(0, 1)) * t;
        } else {
            int i = 0;
            if (m.coeff(1, 1) > m.coeff(0, 0)) i = 1;
            if (m.coeff(2, 2) > m.coeff(i, i)) i = 2;
            const int j = (i + 1) % 3, k = (j + 1) % 3;
            t = sqrt(m.coeff(i, i) - m.coeff(j, j) - m.coeff(k, k) + S(1));
            S qv[3]; qv[i] = S(0.5) * t; t = S(0.5) / t;
            w_ = (m.coeff(k, j) - m.coeff(j, k)) * t; qv[j] = (m.coeff(j, i) + m.coeff(i, j)) * t; qv[k] = (m.coeff(k, i) + m.coeff(i, k)) * t;
            x_ = qv[0]; y_ = qv[1]; z_ = qv[2];
        }
    }
    Quaternion& setIdentity() { w_ = S(1); x_ = y_ = z_ = S(0); return *this; }
    static Quaternion Identity() { return Quaternion(S(1), S(0), S(0), S(0)); }
    template <class A, class B> static Quaternion FromTwoVectors(const MatrixBase<A>& a, const MatrixBase<B>& b) {
        using std::sqrt;
        const Vector3 v0 = Vector3(a.vget(0), a.vget(1), a.vget(2)).normalized(), v1 = Vector3(b.vget(0), b.vget(1), b.vget(2)).normalized();
        const S c = v0.dot(v1);
        if (c < S(-1) + S(1e-12)) {                                   // opposite vectors: any axis orthogonal to v0
            Vector3 ax = Vector3(S(1), S(0), S(0)).cross(v0);
            if (ax.squaredNorm() < S(1e-12)) ax = Vector3(S(0), S(1), S(0)).cross(v0);
            ax.normalize();
            return Quaternion(S(0), ax(0), ax(1), ax(2));
        }
        const Vector3 ax = v0.cross(v1);
        const S s = sqrt((S(1) + c) * S(2)), invs = S(1) / s;
        return Quaternion(s * S(0.5), ax(0) * invs, ax(1) * invs, ax(2) * invs);
    }
    template <class T> Quaternion<T> cast() const { return Quaternion<T>(T(w_), T(x_), T(y_), T(z_)); }
    Quaternion slerp(const S& t, const Quaternion& o) const {
        using std::acos; using std::sin; using std::abs;
        const S d = dot(o), ad = abs(d);
        S s0, s1;
        if (ad >= S(1) - S(1e-15)) { s0 = S(1) - t; s1 = t; }
        else { const S th = acos(ad), st = sin(th); s0 = sin((S(1) - t) * th) / st; s1 = sin(t * th) / st; }
        if (d < S(0)) s1 = -s1;
        return Quaternion(s0 * w_ + s1 * o.w_, s0 * x_ + s1 * o.x_, s0 * y_ + s1 * o.y_, s0 * z_ + s1 * o.z_);
    }
    S angularDistance(const Quaternion& o) const { using std::atan2; using std::abs; const Quaternion d = *this * o.conjugate(); return S(2) * atan2(d.vec().norm(), abs(d.w())); }
  private:
    S w_, x_, y_, z_;
};
typedef Quaternion<double> Quaterniond; typedef Quaternion<float> Quaternionf;

// ------------------------------------------------------------------------------------------------ AngleAxis (parsed by templates of the reference that are never instantiated here)
template <class S> class AngleAxis {
  public:
    typedef Matrix<S, 3, 1> Vector3;
    AngleAxis() : angle_(S(0)), axis_(Vector3(S(1), S(0), S(0))) {}
    template <class O> AngleAxis(const S& angle, const MatrixBase<O>& axis) : angle_(angle), axis_(axis) {}
    template <class O> explicit AngleAxis(const MatrixBase<O>& R) { fromRotationMatrix(R); }
    explicit AngleAxis(const Quaternion<S>& q) { *this = q; }
    AngleAxis& operator=(const Quaternion<S>& q) {
        using std::atan2;
        S n = q.vec().norm();
        if (n > S(0)) { angle_ = S(2) * atan2(n, q.w() < S(0) ? -q.w() : q.w()); if (q.w() < S(0)) n = -n; axis_ = q.vec() / n; }
        else { angle_ = S(0); axis_ = Vector3(S(1), S(0), S(0)); }
        return *this;
    }
    template <class O> AngleAxis& fromRotationMatrix(const MatrixBase<O>& R) { Quaternion<S> q; q.fromRotationMatrix(R); return *this = q; }
    const S& angle() const { return angle_; } S& angle() { return angle_; }
    const Vector3& axis() const { return axis_; } Vector3& axis() { return axis_; }
    Matrix<S, 3, 3> toRotationMatrix() const { return Quaternion<S>(*this).toRotationMatrix(); }
    Quaternion<S> operator*(const AngleAxis& o) const { return Quaternion<S>(*this) * Quaternion<S>(o); }
    Quaternion<S> operator*(const Quaternion<S>& o) const { return Quaternion<S>(*this) * o; }
  private:
    S angle_; Vector3 axis_;
};
template <class S> Quaternion<S> operator*(const Quaternion<S>& q, const AngleAxis<S>& a) { return q * Quaternion<S>(a); }
typedef AngleAxis<double> AngleAxisd; typedef AngleAxis<float> AngleAxisf;

template <class T> struct NumTraits { static T epsilon() { return std::numeric_limits<T>::epsilon(); } static T highest() { return std::numeric_limits<T>::max(); } };

}  // namespace Eigen
#endif
