// LDS-DMA semantics probe (gfx950): where do the bytes of global_load_lds_dwordx4 land, full and half wave?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(1))) const void* gptr;
typedef __attribute__((address_space(3))) void* lptr;
__global__ void probe(const unsigned* src, unsigned* out) {
    __shared__ __attribute__((aligned(16))) unsigned lds[1024];
    const int lane = threadIdx.x;
    for (int i = lane; i < 1024; i += 64) lds[i] = 0xdeadbeefu;
    __syncthreads();
    __builtin_amdgcn_global_load_lds((gptr)(src + 4 * lane), (lptr)lds, 16, 0, 0);
    if (lane < 32) __builtin_amdgcn_global_load_lds((gptr)(src + 1000 + 4 * lane), (lptr)(lds + 512), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 1024; i += 64) out[i] = lds[i];
}
int main() {
    std::vector<unsigned> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = i;
    unsigned *d, *o;
    hipMalloc(&d, 4096 * 4); hipMalloc(&o, 1024 * 4);
    hipMemcpy(d, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o);
    std::vector<unsigned> r(1024);
    hipMemcpy(r.data(), o, 1024 * 4, hipMemcpyDeviceToHost);
    printf("full wave: ");
    for (int i = 0; i < 24; ++i) printf("%x ", r[i]);
    printf("... [252..259] ");
    for (int i = 252; i < 260; ++i) printf("%x ", r[i]);
    printf("\nhalf wave @512: ");
    for (int i = 512; i < 532; ++i) printf("%x ", r[i]);
    printf("... [636..644] ");
    for (int i = 636; i < 644; ++i) printf("%x ", r[i]);
    printf("\n");
    int bad = 0;
    for (int i = 0; i < 256; ++i) bad += r[i] != (unsigned)i;
    for (int i = 0; i < 128; ++i) bad += r[512 + i] != (unsigned)(1000 + i);
    printf("lane-linear layout mismatches: %d\n", bad);
    return 0;
}
