// accuracy of v_rsq_f64 (the hardware estimate behind pivot_rsqrt, solver_kernels.hip) and of the refinements built on it:
//   two Newton steps (what the chain kernels use) vs ONE third-order step y0 (1 + e/2 + 3 e^2 / 8), e = 1 - d y0^2
// hipcc --offload-arch=gfx950 -O3 rsq_accuracy.hip -o rsq_accuracy && ./rsq_accuracy
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
__global__ void k(const double* d, double* y0, double* y2, double* y3, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = d[i];
    const double a = __builtin_amdgcn_rsq(x);
    y0[i] = a;
    const double e = fma(-x * a, a, 1.0);
    const double b = fma(0.5 * a, e, a);
    const double e1 = fma(-x * b, b, 1.0);
    y2[i] = fma(0.5 * b, e1, b);
    y3[i] = fma(a * e, fma(0.375, e, 0.5), a);
}
int main() {
    const int n = 1 << 22;
    std::mt19937_64 g(7);
    std::vector<double> h(n);
    for (int i = 0; i < n; ++i) { const double u = std::generate_canonical<double, 53>(g); const int ex = (int)(g() % 80) - 40; h[i] = std::ldexp(1.0 + u, ex); }
    double *d, *a, *b, *c;
    hipMalloc(&d, n * 8); hipMalloc(&a, n * 8); hipMalloc(&b, n * 8); hipMalloc(&c, n * 8);
    hipMemcpy(d, h.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, a, b, c, n);
    std::vector<double> ha(n), hb(n), hc(n);
    hipMemcpy(ha.data(), a, n * 8, hipMemcpyDeviceToHost); hipMemcpy(hb.data(), b, n * 8, hipMemcpyDeviceToHost); hipMemcpy(hc.data(), c, n * 8, hipMemcpyDeviceToHost);
    double m0 = 0, m2 = 0, m3 = 0; long diff23 = 0;
    for (int i = 0; i < n; ++i) {
        const long double ex = 1.0L / sqrtl((long double)h[i]);
        m0 = std::fmax(m0, (double)fabsl((ha[i] - ex) / ex)); m2 = std::fmax(m2, (double)fabsl((hb[i] - ex) / ex)); m3 = std::fmax(m3, (double)fabsl((hc[i] - ex) / ex));
        diff23 += hb[i] != hc[i];
    }
    printf("max relative error: v_rsq_f64 %.3g (2^%.1f), two Newton steps %.3g (%.2f ulp), one third-order step %.3g (%.2f ulp); results differ in %ld of %d\n",
           m0, std::log2(m0), m2, m2 / 1.11e-16, m3, m3 / 1.11e-16, diff23, n);
    return 0;
}
