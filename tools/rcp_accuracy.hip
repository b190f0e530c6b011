#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double* x, double* r0, double* r1, double* r2, double* r3, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = x[i];
    double r = __builtin_amdgcn_rcp(v);
    r0[i] = r;
    r = fma(fma(-v, r, 1.0), r, r);
    r1[i] = r;
    r = fma(fma(-v, r, 1.0), r, r);
    r2[i] = r;
    // cubic (Halley-type) step from the hardware estimate: r0 (1 + e + e^2), e = 1 - v r0: three FMAs instead of four
    const double q = __builtin_amdgcn_rcp(v);
    const double e = fma(-v, q, 1.0);
    r3[i] = fma(q, fma(e, e, e), q);
}
int main() {
    const int n = 1 << 20;
    std::vector<double> x(n), a(n), b(n), c(n), d(n);
    srand(1);
    for (int i = 0; i < n; ++i) x[i] = ldexp(1.0 + (double)rand() / RAND_MAX, (rand() % 60) - 30) * ((rand() & 1) ? 1 : -1);
    double *dx, *d0, *d1, *d2, *d3;
    hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8); hipMalloc(&d3, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, d0, d1, d2, d3, n);
    hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(c.data(), d2, n * 8, hipMemcpyDeviceToHost); hipMemcpy(d.data(), d3, n * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0;
    for (int i = 0; i < n; ++i) {
        long double t = 1.0L / (long double)x[i];
        e0 = fmax(e0, (double)fabsl(((long double)a[i] - t) / t));
        e1 = fmax(e1, (double)fabsl(((long double)b[i] - t) / t));
        e2 = fmax(e2, (double)fabsl(((long double)c[i] - t) / t));
    }
    double e3 = 0; int n2 = 0, n3 = 0, n23 = 0;
    for (int i = 0; i < n; ++i) {
        long double t = 1.0L / (long double)x[i];
        e3 = fmax(e3, (double)fabsl(((long double)d[i] - t) / t));
        const double cr = 1.0 / x[i];
        n2 += c[i] != cr; n3 += d[i] != cr; n23 += c[i] != d[i];
    }
    printf("cubic step %.3e (2^%.1f); differs from correctly rounded 1/x: +2 newton %d, cubic %d of %d; the two differ in %d\n", e3, log2(e3), n2, n3, n, n23);
    printf("max rel err: rcp %.3e (2^%.1f)  +1 newton %.3e (2^%.1f)  +2 newton %.3e (2^%.1f)\n", e0, log2(e0), e1, log2(e1), e2, log2(e2));
    return 0;
}
