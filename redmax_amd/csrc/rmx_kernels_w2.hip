// rmx_kernels_w2.hip -- the two-wave BDF1 step kernel for trees of 33..64 nodes (rmx_device.h, "two wavefronts per trajectory").
//
// A trajectory gets a workgroup of TWO wavefronts.  Wave 0 owns the rollout (state, front, line search, every decision); the
// second wave is a linear-algebra helper: for every (g,H) solve wave 0 stages the operands of H in LDS and posts a command, each
// wave assembles the half of H whose columns c = 2 t + W it owns on the matrix cores, straight into the row-major staging the
// block-column solve eliminates in (lu_solve_neg_diag64), and they share that elimination (rmx_device.h, "two wavefronts").
// Chosen by the launcher while a GPU holds at most one rollout per CU (launch_step_np_64; RMX_W2=0 / 1 forces the choice).
// LDS of a workgroup: [per-node constants][exchange area: the command word][one scratch area: the front's scans / the operand
// staging / H].  Wave 0's evaluation stages order their LDS traffic wave-locally (RMX_SYNC); workgroup barriers only at the
// hand-overs of a solve (B1 command, B2 operands consumed, B3 H complete, one per phase 0..2).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ double* rmx_smem_base() {
    extern __shared__ __attribute__((aligned(16))) double rmx_smem[];
    return rmx_smem;
}
// wave-local ordering of LDS traffic: wait for this wave's outstanding LDS operations (an LDS queue serves one wave in order), no s_barrier
#define RMX_SYNC()                                                \
    do {                                                          \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   \
        __builtin_amdgcn_wave_barrier();                          \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   \
    } while (0)
#define RMX_CONSTS(sAcc, n, NP) (rmx_smem_base())

#include "rmx_host.h"

// pivot-only Newton of one step (the pivot policy's back-off, or lu_mode = 1): the one-wave code, a real call (see w2_pivoted_solve)
template <int NP>
__device__ __attribute__((noinline)) double w2_pivot_only_newton(const DevModel& M, const DevOpts& o, double* sAcc, const int lane, const double xg,
                                                                 const double q0, NodeOut& last, int& iters, int& halv, int& status, PivotPolicy& piv) {
    return newton_impl<NP, true, false>(M, o, sAcc, nullptr, lane, xg, q0, xg, o.h, last, iters, halv, status, piv);
}

// Wave 0: simLoop (driverRedMaxBDF1.m:57-91) exactly as k_step_bdf1, with the (g,H) solves of the fast Newton shared with the helper wave.
template <int NP, bool PROF>
__device__ __forceinline__ void step_bdf1_w2_owner(const DevModel& M, const DevOpts& o, const StepArgs& a, double* sAcc, double* sX, unsigned long long* dprof) {
    unsigned long long prof[4] = {0ull, 0ull, 0ull, 0ull};
    const int lane = threadIdx.x & 63, traj = blockIdx.x;
    const int id = (lane < M.n) ? M.idx[lane] : -1;
    const size_t off = (size_t)traj * M.nr + (id >= 0 ? id : 0);
    double q = id >= 0 ? a.q[off] : 0.0;
    double qd = id >= 0 ? a.qd[off] : 0.0;
    int iters = 0, halv = 0, status = 0;
    PivotPolicy piv;
    for (int s = 0; s < a.nsteps; ++s) {
        const double q0 = q, qd0 = qd;
        const double xg = q0 + o.h * qd0;
        NodeOut last;
        double x;
        if (o.lu_mode != 0 || piv.hold > 0) {      // wave-uniform: this step on the pivot-only Newton, alone (the helper keeps waiting)
            if (piv.hold > 0) --piv.hold;
            // An out-of-line call takes COPIES of everything it gets by reference: an object whose address escapes lives in
            // scratch for the whole kernel, and for the model constants that turned every M.n / M.rounds / M.grav of the inlined
            // front into a scratch load (+ s_waitcnt vmcnt(0)): 21.5 k instead of 13.5 k cycles per evaluation.
            const DevModel Mc = M;
            const DevOpts oc = o;
            NodeOut l2;
            int it2 = 0, hv2 = 0, st2 = 0;
            PivotPolicy pv2 = piv;
            x = w2_pivot_only_newton<NP>(Mc, oc, sAcc, lane, xg, q0, l2, it2, hv2, st2, pv2);
            last = l2;
            iters += it2;
            halv += hv2;
            status |= st2;
            piv = pv2;
        } else {
            x = newton_w2<NP, PROF>(M, o, sAcc, sX, lane, xg, q0, xg, o.h, last, iters, halv, status, piv, prof);
            pivot_policy_update(piv);
        }
        qd = (x - q0) / o.h;
        q = x;
        if (a.histT) {
            const double T = wave_sum(last.eT), V = wave_sum(last.eV);
            if (lane == 0) {
                a.histT[(size_t)s * a.B + traj] = T;
                a.histV[(size_t)s * a.B + traj] = V;
            }
        }
        if (a.histQ && id >= 0) {
            a.histQ[(size_t)s * a.B * M.nr + off] = q;
            a.histQd[(size_t)s * a.B * M.nr + off] = qd;
        }
    }
    if (lane == 0) sX[0] = 0.0;            // command: exit
    __syncthreads();                       // B1 of the helper's last wait
    if (PROF && lane == 0) {
        for (int c = 0; c < 3; ++c) dprof[(size_t)(2 * traj) * 4 + c] = prof[c];
        dprof[(size_t)(2 * traj) * 4 + 3] = (unsigned long long)iters;
    }
    if (id >= 0) {
        a.q[off] = q;
        a.qd[off] = qd;
    }
    if (lane == 0 && a.it) {
        a.it[traj] += iters;
        a.ls[traj] += halv;
        a.status[traj] |= status;
    }
}

template <int NP, bool PROF>
__global__ void __launch_bounds__(128) k_step_bdf1_w2(const DevModel M, const DevOpts o, const StepArgs a, unsigned long long* dprof) {
    double* smem = rmx_smem_base();
    constexpr int CS = cstride(NP);
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double* sX = smem + NCONST * CS;
    double* sAcc0 = sX + W2_XCH;                       // the front's scratch / operand staging / H (shared by the two waves)
    // per-node constants (the layout eval_front_e2 reads; see smem_setup in rmx_kernels.hip)
    if (threadIdx.x < CS) {
        const int j = threadIdx.x;
        const bool in = j < M.n;
        double* c = smem;
        for (int r = 0; r < 36; ++r) c[r * CS + j] = in ? M.K[r * MAXN + j] : ((r == 0 || r == 4 || r == 8) ? 1.0 : 0.0);
        c += 36 * CS;
        for (int r = 0; r < 6; ++r) c[r * CS + j] = in ? M.sb[r * MAXN + j] : 0.0;
        c += 6 * CS;
        for (int r = 0; r < 4; ++r) c[r * CS + j] = in ? M.I4[r * MAXN + j] : 0.0;
        c += 4 * CS;
        for (int r = 0; r < 8; ++r) c[r * CS + j] = in ? M.prm[r * MAXN + j] : 0.0;
        c += 8 * CS;
        c[j] = in ? (double)M.type[j] : 0.0;
        c += CS;
        c[j] = in ? __longlong_as_double((long long)M.rel[j]) : 0.0;
        c[CS + j] = in ? __longlong_as_double((long long)M.rel[MAXN + j]) : 0.0;
        c += 2 * CS;
        for (int r = 0; r < MAXROUNDS; ++r) c[r * CS + j] = in ? (double)M.anc[r * MAXN + j] : -1.0;
        c += MAXROUNDS * CS;
        c[j] = in ? (double)M.end[j] : (double)M.n;
        c += CS;
        for (int r = 0; r < 4; ++r) c[r * CS + j] = 0.0;
    }
    if (threadIdx.x < ACC_STRIDE) sAcc0[M.n * ACC_STRIDE + threadIdx.x] = 0.0;     // zero row n of the front's scratch (end-of-tree suffix)
    __syncthreads();
    if (w == 0) {
        step_bdf1_w2_owner<NP, PROF>(M, o, a, sAcc0, sX, dprof);
    } else {
        unsigned long long prof[4] = {0ull, 0ull, 0ull, 0ull};
        w2_helper_loop<NP, PROF>(M, sAcc0, sX, lane, prof);
        if (PROF && lane == 0)
            for (int c = 0; c < 4; ++c) dprof[(size_t)(2 * blockIdx.x + 1) * 4 + c] = prof[c];
    }
}

size_t rmx_w2_smem_bytes(const rmx_model* m) {
    return sizeof(double) * ((size_t)NCONST * cstride(64) + W2_XCH + (size_t)w2_acc_doubles(m->n));
}

void launch_step_w2_64(const rmx_model* m, const rmx_batch* b, const DevOpts& o, const StepArgs& a) {
    const dim3 grid(b->B), block(128);
    const size_t bytes = rmx_w2_smem_bytes(m);
    // 64 KiB+ of dynamic LDS needs the opt-in; it is a per-device attribute and costs microseconds, so it is set at every launch
    // rather than cached (a process may drive several devices from several threads)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_step_bdf1_w2<64, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (getenv("RMX_W2_PROF")) {   // development aid: shader-clock cycles of (front, Hessian, elimination) per wave, printed to stderr
        unsigned long long* d = nullptr;
        const size_t nb = sizeof(unsigned long long) * 8 * (size_t)b->B;
        if (hipMalloc((void**)&d, nb) != hipSuccess) return;
        (void)hipMemsetAsync(d, 0, nb, b->stream);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_step_bdf1_w2<64, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        k_step_bdf1_w2<64, true><<<grid, block, bytes, b->stream>>>(m->dm, o, a, d);
        std::vector<unsigned long long> h(8 * (size_t)b->B);
        (void)hipMemcpyAsync(h.data(), d, nb, hipMemcpyDeviceToHost, b->stream);
        (void)hipStreamSynchronize(b->stream);
        (void)hipFree(d);
        double acc[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
        for (int t = 0; t < b->B; ++t)
            for (int w = 0; w < 2; ++w)
                for (int c = 0; c < 4; ++c) acc[w][c] += (double)h[(size_t)(2 * t + w) * 4 + c];
        for (int w = 0; w < 2; ++w)
            fprintf(stderr, "w2 prof wave %d: per Newton iteration front %.0f  hess %.0f  elimination %.0f cycles (%.0f iterations per rollout)\n", w,
                    acc[w][0] / acc[w][3], acc[w][1] / acc[w][3], acc[w][2] / acc[w][3], acc[w][3] / b->B);
        return;
    }
    k_step_bdf1_w2<64, false><<<grid, block, bytes, b->stream>>>(m->dm, o, a, nullptr);
}
