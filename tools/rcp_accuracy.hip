#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double* x, double* r0, double* r1, double* r2, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = x[i];
    double r = __builtin_amdgcn_rcp(v);
    r0[i] = r;
    r = fma(fma(-v, r, 1.0), r, r);
    r1[i] = r;
    r = fma(fma(-v, r, 1.0), r, r);
    r2[i] = r;
}
int main() {
    const int n = 1 << 20;
    std::vector<double> x(n), a(n), b(n), c(n);
    srand(1);
    for (int i = 0; i < n; ++i) x[i] = ldexp(1.0 + (double)rand() / RAND_MAX, (rand() % 60) - 30) * ((rand() & 1) ? 1 : -1);
    double *dx, *d0, *d1, *d2;
    hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, d0, d1, d2, n);
    hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(c.data(), d2, n * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0;
    for (int i = 0; i < n; ++i) {
        long double t = 1.0L / (long double)x[i];
        e0 = fmax(e0, (double)fabsl(((long double)a[i] - t) / t));
        e1 = fmax(e1, (double)fabsl(((long double)b[i] - t) / t));
        e2 = fmax(e2, (double)fabsl(((long double)c[i] - t) / t));
    }
    printf("max rel err: rcp %.3e (2^%.1f)  +1 newton %.3e (2^%.1f)  +2 newton %.3e (2^%.1f)\n", e0, log2(e0), e1, log2(e1), e2, log2(e2));
    return 0;
}
