// Calibrates s_memtime against wall time (HIP events): prints ticks per microsecond, idle chip and under fp64 load.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(unsigned long long ticks, double* out) {
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    double a = threadIdx.x * 1e-3, b = 1.0000001;
    while (__builtin_amdgcn_s_memtime() - t0 < ticks) {
#pragma unroll
        for (int i = 0; i < 64; ++i) a = fma(a, b, 1e-9);
    }
    out[blockIdx.x * 64 + threadIdx.x] = a;
}
int main() {
    double* out; hipMalloc(&out, sizeof(double) * 64 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int nblk : {1, 1024}) {
        for (int rep = 0; rep < 2; ++rep) {
            const unsigned long long ticks = 200000000ull;
            hipEventRecord(e0); spin<<<nblk, 64>>>(ticks, out); hipEventRecord(e1); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("blocks %4d: %llu ticks in %.3f ms -> %.1f ticks/us\n", nblk, ticks, ms, ticks / (ms * 1e3));
        }
    }
    return 0;
}
