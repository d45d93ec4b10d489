#include <hip/hip_runtime.h>
#include <cstdio>
template <int C> __device__ __forceinline__ void fmac_bcast(double& acc, const double x, const double nl) {
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(nl), "n"(C));
}
__global__ void k(const double* in, double* out) {
    const int lane = threadIdx.x;
    double x = in[lane], nl = in[64 + lane], acc = in[128 + lane];
    asm volatile("s_nop 1");
    fmac_bcast<3>(acc, x, nl);
    fmac_bcast<7>(acc, x, nl);
    out[lane] = acc;
}
int main() {
    double h[192], o[64];
    for (int i = 0; i < 192; ++i) h[i] = 0.001 * (i * 37 % 101) + 1.0;
    double *d, *e; hipMalloc(&d, sizeof h); hipMalloc(&e, sizeof o);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, e);
    hipMemcpy(o, e, sizeof o, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const int row = l & ~15;
        double want = h[128 + l];
        want = __builtin_fma(h[row + 3], h[64 + l], want);
        want = __builtin_fma(h[row + 7], h[64 + l], want);
        if (want != o[l]) { ++bad; if (bad < 4) printf("lane %d got %.17g want %.17g\n", l, o[l], want); }
    }
    printf("bad %d\n", bad);
    return 0;
}
