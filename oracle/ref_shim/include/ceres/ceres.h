// ceres/ceres.h stand-in (oracle/ref_shim, TEST INFRASTRUCTURE): the modelling API of Ceres 1.14 that the reference's factor
// layer derives from -- CostFunction / SizedCostFunction / AutoDiffCostFunction (Jets seeded block by block, one pass),
// LossFunction / HuberLoss, LocalParameterization / QuaternionParameterization -- restated from the API reference bundled with
// the reference (GraphGNSSLibV1.1/docs/source/nnls_modeling.rst:75-140 CostFunction, :166-262 Sized/AutoDiff, :1000-1063 loss,
// :1312-1327 quaternion plus).  No Problem, no Solver: the solve loop is NOT pinned by this library.
#pragma once
#include <algorithm>
#include <cmath>
#include <limits>
#include <utility>
#include <vector>
#include "ceres/jet.h"
namespace ceres {
typedef int int32;
enum Ownership { DO_NOT_TAKE_OWNERSHIP, TAKE_OWNERSHIP };
const int DYNAMIC = -1;

class CostFunction {
  public:
    CostFunction() : num_residuals_(0) {}
    virtual ~CostFunction() {}
    virtual bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const = 0;
    const std::vector<int32>& parameter_block_sizes() const { return parameter_block_sizes_; }
    int num_residuals() const { return num_residuals_; }
  protected:
    std::vector<int32>* mutable_parameter_block_sizes() { return &parameter_block_sizes_; }
    void set_num_residuals(int n) { num_residuals_ = n; }
  private:
    std::vector<int32> parameter_block_sizes_;
    int num_residuals_;
};

template <int kNumResiduals, int... Ns> class SizedCostFunction : public CostFunction {
  public:
    SizedCostFunction() {
        set_num_residuals(kNumResiduals);
        const int sizes[] = {Ns...};
        for (int s : sizes) if (s > 0) mutable_parameter_block_sizes()->push_back(s);       // (trailing zeros = unused slots of the 1.14 signature)
    }
    virtual ~SizedCostFunction() {}
};

namespace internal {
template <int... Ns> struct Sum;
template <> struct Sum<> { static const int value = 0; };
template <int N, int... Ns> struct Sum<N, Ns...> { static const int value = N + Sum<Ns...>::value; };
template <class Functor, class T, std::size_t... I> inline bool call_functor(const Functor& f, T const* const* p, T* out, std::index_sequence<I...>) { return f(p[I]..., out); }
}  // namespace internal

// Jacobians by Jets: ONE evaluation with Jet<double, N0 + N1 + ...>, block i seeded in the dual parts [off_i, off_i + N_i)
// (autodiff.h of 1.14 does the same when the total fits its stack budget).  jacobians[i] is row-major kNumResiduals x N_i.
template <class CostFunctor, int kNumResiduals, int... Ns> class AutoDiffCostFunction : public SizedCostFunction<kNumResiduals, Ns...> {
  public:
    explicit AutoDiffCostFunction(CostFunctor* functor) : functor_(functor) {}
    virtual ~AutoDiffCostFunction() { delete functor_; }
    virtual bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const {
        const int nb = sizeof...(Ns);
        if (!jacobians) return internal::call_functor(*functor_, parameters, residuals, std::make_index_sequence<sizeof...(Ns)>());
        const int sizes[] = {Ns...};
        const int NT = internal::Sum<Ns...>::value;
        typedef Jet<double, internal::Sum<Ns...>::value> JetT;
        std::vector<JetT> x((size_t)NT), out((size_t)kNumResiduals);
        const JetT* ptrs[sizeof...(Ns)];
        int off = 0;
        for (int b = 0; b < nb; ++b) {
            ptrs[b] = x.data() + off;
            for (int k = 0; k < sizes[b]; ++k) x[off + k] = JetT(parameters[b][k], off + k);
            off += sizes[b];
        }
        if (!internal::call_functor(*functor_, ptrs, out.data(), std::make_index_sequence<sizeof...(Ns)>())) return false;
        for (int r = 0; r < kNumResiduals; ++r) residuals[r] = out[r].a;
        off = 0;
        for (int b = 0; b < nb; ++b) {
            if (jacobians[b]) for (int r = 0; r < kNumResiduals; ++r) for (int k = 0; k < sizes[b]; ++k) jacobians[b][r * sizes[b] + k] = out[r].v[off + k];
            off += sizes[b];
        }
        return true;
    }
  private:
    CostFunctor* functor_;
};

class LossFunction {
  public:
    virtual ~LossFunction() {}
    virtual void Evaluate(double sq_norm, double out[3]) const = 0;
};
class TrivialLoss : public LossFunction { public: virtual void Evaluate(double s, double rho[3]) const { rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; } };
// rho(s) = s for s <= a^2, 2 a sqrt(s) - a^2 beyond (nnls_modeling.rst:1044-1049; loss_function.cc of 1.14 clamps rho' from below)
class HuberLoss : public LossFunction {
  public:
    explicit HuberLoss(double a) : a_(a), b_(a * a) {}
    virtual void Evaluate(double s, double rho[3]) const {
        if (s > b_) { const double r = std::sqrt(s); rho[0] = 2.0 * a_ * r - b_; rho[1] = std::max(std::numeric_limits<double>::min(), a_ / r); rho[2] = -rho[1] / (2.0 * s); }
        else { rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; }
    }
  private:
    const double a_, b_;
};
class CauchyLoss : public LossFunction {
  public:
    explicit CauchyLoss(double a) : b_(a * a), c_(1 / b_) {}
    virtual void Evaluate(double s, double rho[3]) const { const double sum = 1.0 + s * c_, inv = 1.0 / sum; rho[0] = b_ * std::log(sum); rho[1] = std::max(std::numeric_limits<double>::min(), inv); rho[2] = -c_ * (inv * inv); }
  private:
    const double b_, c_;
};

class LocalParameterization {
  public:
    virtual ~LocalParameterization() {}
    virtual bool Plus(const double* x, const double* delta, double* x_plus_delta) const = 0;
    virtual bool ComputeJacobian(const double* x, double* jacobian) const = 0;       // row-major GlobalSize x LocalSize
    virtual int GlobalSize() const = 0;
    virtual int LocalSize() const = 0;
};
// x (+) delta = [cos|d|, sin|d| d/|d|] * x, (w, x, y, z) order (nnls_modeling.rst:1312-1327; local_parameterization.cc of 1.14)
class QuaternionParameterization : public LocalParameterization {
  public:
    virtual bool Plus(const double* x, const double* delta, double* x_plus_delta) const {
        const double norm_delta = std::sqrt(delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2]);
        if (norm_delta > 0.0) {
            const double sin_delta_by_delta = std::sin(norm_delta) / norm_delta;
            const double q[4] = {std::cos(norm_delta), sin_delta_by_delta * delta[0], sin_delta_by_delta * delta[1], sin_delta_by_delta * delta[2]};
            x_plus_delta[0] = q[0] * x[0] - q[1] * x[1] - q[2] * x[2] - q[3] * x[3];
            x_plus_delta[1] = q[0] * x[1] + q[1] * x[0] + q[2] * x[3] - q[3] * x[2];
            x_plus_delta[2] = q[0] * x[2] - q[1] * x[3] + q[2] * x[0] + q[3] * x[1];
            x_plus_delta[3] = q[0] * x[3] + q[1] * x[2] - q[2] * x[1] + q[3] * x[0];
        } else for (int i = 0; i < 4; ++i) x_plus_delta[i] = x[i];
        return true;
    }
    virtual bool ComputeJacobian(const double* x, double* j) const {
        j[0] = -x[1]; j[1] = -x[2]; j[2] = -x[3];
        j[3] = x[0];  j[4] = x[3];  j[5] = -x[2];
        j[6] = -x[3]; j[7] = x[0];  j[8] = x[1];
        j[9] = x[2];  j[10] = -x[1]; j[11] = x[0];
        return true;
    }
    virtual int GlobalSize() const { return 4; }
    virtual int LocalSize() const { return 3; }
};
}  // namespace ceres
