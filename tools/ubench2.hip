// Issue cost of instruction MIXES for a lone wavefront per SIMD (round 3): what does a non-fp64 instruction cost when it sits between
// fp64 VALU instructions?  Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/ubench2.hip -o /tmp/ubench2 && /tmp/ubench2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define FMA(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c0), "v"(c1));
#define R16(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)
#define M1(i) FMA(i) asm volatile("s_or_b32 %0, %0, 1" : "+s"(sacc));
#define M2(i) FMA(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(iv[i]) : "v"(lane));
#define M3(i) asm volatile("v_mov_b32 %0, %1" : "=v"(iv[i]) : "v"(lane));
#define M4(i) FMA(i) asm volatile("v_accvgpr_read_b32 %0, a0" : "=v"(iv[i]));
#define M5(i) FMA(i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(ld[i]) : "v"(sa), "n"(i * 512));
#define M6(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[0]) : "v"(c0), "v"(c1));
#define M7(i) FMA(i) asm volatile("s_nop 0");
#define M8(i) FMA(i) asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(sacc) : "v"(iv[i]));
#define M9(i) FMA(i) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(iv[i]) : "v"(lane));
#define M10(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c1)); asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c0));
#define M11(i) FMA(i) asm volatile("s_mov_b64 s[20:21], exec\n\ts_mov_b64 exec, s[20:21]" ::: "s20", "s21");
#define M12(i) FMA(i) asm volatile("s_or_b32 %0, %0, 1\n\ts_or_b32 %0, %0, 2" : "+s"(sacc));
#define M13(i) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[(i + 8) & 15]), "v"(c1));
#define M14(i) FMA(i) asm volatile("v_cmp_lt_i32 vcc, %2, %3\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_cndmask_b32 %1, %1, %0, vcc" : "+v"(iv[i]), "+v"(iv[(i + 1) & 15]) : "v"(lane), "v"(lane2) : "vcc");

template <int MODE>
__global__ void __launch_bounds__(64) k(double* out, unsigned long long* cyc, int reps) {
    const int lane = threadIdx.x;
    const int lane2 = lane ^ 5;
    double a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = 1.0 + 1e-3 * (lane + i);
    const double c0 = 0.999, c1 = 1e-6 * lane;
    int iv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) iv[i] = lane + i;
    int sacc = reps;
    __shared__ double sm[64 * 17];
#pragma unroll
    for (int i = 0; i < 17; ++i) sm[lane * 17 + i] = lane + i;
    __syncthreads();
    const unsigned sa = (unsigned)(lane * 8);
    double ld[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) ld[i] = 0.0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
        if (MODE == 0) { R16(FMA) }
        if (MODE == 1) { R16(M1) }
        if (MODE == 2) { R16(M2) }
        if (MODE == 3) { R16(M3) }
        if (MODE == 4) { R16(M4) }
        if (MODE == 5) { R16(M5) asm volatile("s_waitcnt lgkmcnt(0)"); }
        if (MODE == 6) { R16(M6) }
        if (MODE == 7) { R16(M7) }
        if (MODE == 8) { R16(M8) }
        if (MODE == 9) { R16(M9) }
        if (MODE == 10) { R16(M10) }
        if (MODE == 11) { R16(M11) }
        if (MODE == 12) { R16(M12) }
        if (MODE == 13) { R16(M13) }
        if (MODE == 14) { R16(M14) }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = sacc;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i] + iv[i] + ld[i];
    out[blockIdx.x * 64 + lane] = s + sm[lane];
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int nblk) {
    double* out; unsigned long long* cyc;
    hipMalloc(&out, sizeof(double) * 64 * nblk); hipMalloc(&cyc, sizeof(unsigned long long) * nblk);
    const int reps = 400;
    k<MODE><<<nblk, 64>>>(out, cyc, reps);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<MODE><<<nblk, 64>>>(out, cyc, reps);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(nblk);
    hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * nblk, hipMemcpyDeviceToHost);
    double m = 0; for (auto v : h) m += v; m /= nblk;
    printf("%-44s blocks %5d: %8.1f memtime-ticks/rep   wall %8.1f ns/rep\n", name, nblk, m / reps, ms * 1e6 / reps);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int nblk : {1024, 2048}) {
        run<0>("16 indep v_fma_f64", nblk);
        run<6>("16 dependent v_fma_f64", nblk);
        run<10>("16 x (v_add_f64 + v_mul_f64)", nblk);
        run<13>("16 x v_fmac_f64_dpp", nblk);
        run<3>("16 x v_mov_b32", nblk);
        run<1>("16 x (fma + s_or_b32)", nblk);
        run<12>("16 x (fma + 2 s_or_b32)", nblk);
        run<11>("16 x (fma + exec save/restore)", nblk);
        run<7>("16 x (fma + s_nop 0)", nblk);
        run<2>("16 x (fma + v_cndmask_b32)", nblk);
        run<14>("16 x (fma + v_cmp + 2 v_cndmask)", nblk);
        run<4>("16 x (fma + v_accvgpr_read)", nblk);
        run<8>("16 x (fma + v_readlane_b32)", nblk);
        run<9>("16 x (fma + v_mov_b32_dpp)", nblk);
        run<5>("16 x (fma + ds_read_b64) + wait", nblk);
    }
    return 0;
}
