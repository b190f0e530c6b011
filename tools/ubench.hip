// Micro-benchmarks of the cross-lane primitives the step kernel leans on, one wavefront per SIMD (1024 blocks of 64).
// Build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o /tmp/ubench && /tmp/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ double readlane_d(double v, int l) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, l);
    hi = __builtin_amdgcn_readlane(hi, l);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }

template <int MODE>
__global__ void __launch_bounds__(64) k(double* out, unsigned long long* cyc, int reps, int dynlane) {
    const int lane = threadIdx.x;
    double a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = 1.0 + 1e-3 * (lane + i);
    double l = 1e-6 * lane;
    __shared__ double sm[64 * 17];
    int pl = dynlane;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
        if (MODE == 0) {   // 16 independent fp64 FMAs
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] = fma(a[i], 0.999, l);
        } else if (MODE == 1) {   // 16 dependent fp64 FMAs
#pragma unroll
            for (int i = 0; i < 16; ++i) l = fma(l, 0.999, a[i]);
        } else if (MODE == 2) {   // 16 x (readlane_d dynamic + FMA) : the LU inner pattern
            pl = __builtin_amdgcn_readfirstlane((pl + 7) & 31);
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] -= l * readlane_d(a[i], pl);
        } else if (MODE == 3) {   // 16 x (readlane_d immediate + FMA)
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] -= l * readlane_d(a[i], 5);
        } else if (MODE == 4) {   // 16 x __shfl (ds_bpermute) of doubles, independent
            const int src = (lane + 1 + r) & 63;
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] += __shfl(a[i], src, 64);
        } else if (MODE == 5) {   // 16 dependent DPP adds (int)
            int v = lane + r;
#pragma unroll
            for (int i = 0; i < 16; ++i) v += dpp_i<0xB1>(v);
            l += v;
        } else if (MODE == 6) {   // LDS: 16 writes (stride 17) then 16 reads transposed
#pragma unroll
            for (int i = 0; i < 16; ++i) sm[lane * 17 + i] = a[i];
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] += sm[((lane + i) & 63) * 17 + i];
            __syncthreads();
        } else if (MODE == 7) {   // dependent chain: ds_bpermute -> add, 16 deep
#pragma unroll
            for (int i = 0; i < 16; ++i) l += __shfl(l, (lane + 1) & 63, 64);
        } else if (MODE == 8) {   // v_rcp_f64 + 2 Newton steps, 4 independent
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                double x = a[i], rr = __builtin_amdgcn_rcp(x);
                rr = fma(fma(-x, rr, 1.0), rr, rr);
                rr = fma(fma(-x, rr, 1.0), rr, rr);
                a[i] = rr + 1.0;
            }
        } else if (MODE == 10) {  // 16 x (ds_swizzle broadcast of lane 5 within each 32-lane half, double) + FMA: 2D-split LU pattern
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(a[i]), 5 << 5);
                const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(a[i]), 5 << 5);
                a[i] -= l * __hiloint2double(hi, lo);
            }
        } else if (MODE == 11) {  // 16 x ds_swizzle broadcast (double) only, accumulate
            double acc = 0.0;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(a[i]), 5 << 5);
                const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(a[i]), 5 << 5);
                acc += __hiloint2double(hi, lo);
            }
            l += acc;
        } else if (MODE == 12) {  // 16 x v_permlane32_swap pair (double) + add
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int lo = __double2loint(a[i]), hi = __double2hiint(a[i]);
                const auto x = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
                const auto y = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
                a[i] += __hiloint2double(y[0], x[0]);
            }
        } else if (MODE == 13) {  // mixed: 8 x (readlane_d + fma) on the VALU, 8 x (ds_swizzle bcast + fma) on the LDS pipe
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(a[i]), 5 << 5);
                const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(a[i]), 5 << 5);
                a[i + 1] -= l * readlane_d(a[i + 1], 5);
                a[i] -= l * __hiloint2double(hi, lo);
            }
        } else if (MODE == 14) {  // 16 x (readlane_d imm + fma), broadcasts batched 8 at a time ahead of their FMAs
#pragma unroll
            for (int h = 0; h < 16; h += 8) {
                double pv[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) pv[i] = readlane_d(a[h + i], 5);
                asm volatile("" ::: "memory");
#pragma unroll
                for (int i = 0; i < 8; ++i) a[h + i] -= l * pv[i];
            }
        } else if (MODE == 15) {   // 16 x (v_mov_b64_dpp row_newbcast + fma): 64-bit DPP broadcast inside each 16-lane row
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] -= l * __builtin_amdgcn_update_dpp(0.0, a[i], 0x150 + 5, 0xF, 0xF, true);
        } else if (MODE == 16) {   // 16 x v_fmac_f64_dpp row_newbcast (broadcast fused into the FMA, inline asm)
            asm volatile("s_nop 1");
#pragma unroll
            for (int i = 0; i < 16; ++i)
                asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[i]), "v"(l));
        } else if (MODE == 17) {   // same, source row distinct from the accumulator (the LU pattern: acc[c] += bcast(piv[c]) * l)
            asm volatile("s_nop 1");
#pragma unroll
            for (int i = 0; i < 8; ++i)
                asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[8 + i]), "v"(l));
#pragma unroll
            for (int i = 0; i < 8; ++i)
                asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:6 row_mask:0xf bank_mask:0xf" : "+v"(a[8 + i]) : "v"(a[i]), "v"(l));
        } else if (MODE == 9) {   // sincos fp64
            double s_, c_;
            sincos(a[0], &s_, &c_);
            a[0] = s_ + c_ * 0.5 + 1.0;
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = l;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * 64 + lane] = s;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int per, int nblk) {
    double* out; unsigned long long* cyc;
    hipMalloc(&out, sizeof(double) * 64 * nblk); hipMalloc(&cyc, sizeof(unsigned long long) * nblk);
    const int reps = 200;
    k<MODE><<<nblk, 64>>>(out, cyc, reps, 3);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<MODE><<<nblk, 64>>>(out, cyc, reps, 3);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(nblk);
    hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * nblk, hipMemcpyDeviceToHost);
    double m = 0; for (auto v : h) m += v; m /= nblk;
    printf("%-44s blocks %5d: %8.1f memtime-ticks/rep  (%6.2f per item)  wall %.3f ms => %.1f ns/rep\n", name, nblk, m / reps, m / reps / per, ms, ms * 1e6 / reps);
    hipFree(out); hipFree(cyc);
}

__global__ void __launch_bounds__(64) dpp_check(double* out) {
    const int lane = threadIdx.x;
    double src = 100.0 + lane, l = 0.5 + lane, acc = 1000.0 * lane;
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(l));
    out[lane] = acc;   // expect 1000*lane + (100 + 16*(lane/16) + 5) * (0.5 + lane)
    out[64 + lane] = __builtin_amdgcn_update_dpp(0.0, src, 0x150 + 9, 0xF, 0xF, true);   // expect 100 + 16*(lane/16) + 9
}

int main() {
    {
        double* d; hipMalloc(&d, sizeof(double) * 128);
        dpp_check<<<1, 64>>>(d);
        double h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l) {
            bad += h[l] != 1000.0 * l + (100.0 + 16 * (l / 16) + 5) * (0.5 + l);
            bad += h[64 + l] != 100.0 + 16 * (l / 16) + 9;
        }
        printf("v_fmac_f64_dpp / v_mov_b64_dpp row_newbcast check: %s (%d mismatches)\n", bad ? "FAIL" : "ok", bad);
        hipFree(d);
    }
    for (int nblk : {1024, 4096}) {
        run<0>("16 indep v_fma_f64", 16, nblk);
        run<1>("16 dependent v_fma_f64", 16, nblk);
        run<2>("16 x (readlane_d dyn + fma)", 16, nblk);
        run<3>("16 x (readlane_d imm + fma)", 16, nblk);
        run<4>("16 indep __shfl double (2 bpermute)", 16, nblk);
        run<5>("16 dependent dpp add (int)", 16, nblk);
        run<6>("LDS 16 wr + sync + 16 rd + sync", 32, nblk);
        run<7>("16 dependent __shfl double + add", 16, nblk);
        run<8>("4 x recip (rcp + 2 newton)", 4, nblk);
        run<9>("sincos fp64", 1, nblk);
        run<10>("16 x (ds_swizzle bcast32 double + fma)", 16, nblk);
        run<11>("16 x ds_swizzle bcast32 double", 16, nblk);
        run<12>("16 x permlane32_swap double + add", 16, nblk);
        run<13>("8 x (readlane+fma) + 8 x (swizzle+fma)", 16, nblk);
        run<14>("16 x (readlane_d imm + fma), batches of 8", 16, nblk);
        run<15>("16 x (v_mov_b64_dpp row_newbcast + fma)", 16, nblk);
        run<16>("16 x v_fmac_f64_dpp row_newbcast (asm)", 16, nblk);
        run<17>("16 x v_fmac_f64_dpp, src != acc (asm)", 16, nblk);
    }
    return 0;
}
