// rmx_big.hip -- the general kernels for trees of 65..BIG_MAXN (256) nodes: ONE WORKGROUP per trajectory, thread = node.
//
// The one-wavefront kernels (rmx_device.h, rmx_kernels.hip) hold a tree in the 64 lanes of a wavefront and stop at 64 nodes, where
// every multi-DOF joint counts one node per DOF (a JointFree3D body is 6 nodes).  The reference has no size limit.  These kernels
// remove the limit with the SAME algebra - the world-frame recursive Newton-Euler with analytic derivatives of DESIGN.md section 3,
// restated node by node in oracle/redmax_tensorfree.c - laid out for a workgroup instead of a wavefront:
//   * root->node path products / sums : pointer jumping over the ancestor table (log2(depth) rounds), the per-node values in LDS
//                                       ([component][n] rows of the dynamic array `dyn`), one workgroup barrier per round
//   * node->leaves subtree sums       : a suffix scan over the depth-first order (Hillis-Steele in place, log2(n) rounds), subtree(j) =
//                                       suffix(j) - suffix(end_j)
//   * the nr x nr Hessian             : thread = row, a loop over the columns (ancestor test j < i < end_j), column-major; in LDS
//                                       when it fits next to the per-node rows (nr <= ~128), else in a global workspace
//   * dx = -H\g                       : guarded elimination on the diagonal first (big_solve_diag: right-looking blocked LU without a
//                                       pivot search - the diagonal block factored in the registers of the wavefront that owns its rows,
//                                       L21 / U12 by substitution against LDS broadcasts, the rank-16 / rank-32 trailing update on the
//                                       fp64 MATRIX CORES (v_mfma_f64_16x16x4_f64: the one contraction of this library that is large
//                                       enough - up to 224 x 224 x 32 per panel), three barriers per panel, one per block in the back
//                                       substitution; H in LDS: 16-column panels in place, H in HBM (nr > ~136): 32-column panels staged
//                                       in LDS).  Every multiplier of the equilibrated matrix is checked against LU_GROWTH_MAX; a solve
//                                       that trips the guard is redone on a re-assembled H with partial pivoting (MATLAB mldivide,
//                                       driverRedMaxBDF1.m:117; first maximum wins; the pivot column scaled by the reciprocal of the
//                                       pivot as dgetf2 does; implicit row permutation, pivot search on DPP butterflies; big_solve /
//                                       big_solve_blocked), rmx_opts.lu_mode = 1 asks for that one always
// Newton (driverRedMaxBDF1.m:94-157) is the reference's, decision for decision, with the stall shortcut and the compensated iterate
// of newton_impl (rmx_device.h), in the rotated form of newton_rot.  Cost (tools/big_tree_bench.py, 256 rollouts, round 4): a 72-link
// chain 0.79 ms per BDF1 step, a 128-link chain 2.07 ms, a 256-link chain 13 - 20 ms (its slowest rollouts do not converge at that
// amplitude: 3.7 ms where all do; 1.02 M ticks per Newton iteration, 2.4 M with the pivot search; 1.9 / 5.4 / 62 ms in round 3); the 64-link chain on the one-wavefront kernels: 0.22 ms.  Covered:
// BDF1, BDF2 (SDIRK2 start), rmx_eval, rmx_energy, histories, JointSpherical / JointFree3D with Euler-chart switching, ground contact
// (ForceGroundCuboid, CT instantiations), rmx_eval_mfd / rmx_compute_values through rmx_eval; not covered: the adjoint, matlab-simple
// Euler (refused by the C ABI for such models).
#include <hip/hip_runtime.h>

#include <type_traits>

// Fused multiply-adds only where the SOURCE writes a * b + c in one expression.  hipcc's default (-ffp-contract=fast) also lets the
// backend fuse a multiply into a later add when the product has no other use - so the residual-only instantiation of big_eval (where
// the Hessian's consumers of a product are compiled out) rounded a few sums differently from the full one, and rmx_eval's g depended on
// whether H was asked for (tests/test_gpu_big_trees.py::test_big_tree_eval_matches_oracle compares the two bit for bit).  These kernels
// are bound by LDS latency, not by the FMA count.
#pragma clang fp contract(on)

#include "rmx_host.h"

namespace {

constexpr int BT = BIG_MAXN;          // threads per workgroup = node slots
#ifndef RMX_BIG_PHASEB_DPP_HL
#define RMX_BIG_PHASEB_DPP_HL 1         // phase B of the guarded LU on DPP broadcasts also when H is in LDS (0: LDS broadcasts there; build variants)
#endif
// How the Newton loop is laid out for the compiler (build variants; ticks per Newton iteration at 72 / 128 / 256 links, profiles/r05p_*):
// one call site for the evaluation with H and one per solve, everything inlined 255 / 394 / 1 065 k; the same with the PIVOTING solve
// out of line (it runs only when the guard trips, and inlined its registers are the hot path's to carry) 229 / 361 / 1 066 k; two call
// sites for the evaluation (guarded path, fallback) 254 / 392 / 1 064 k.  Left to the inliner's own heuristics the layout flipped with
// every change of code size (a Newton loop left out of line costs 60 - 70 k ticks per iteration).
#ifndef RMX_BIG_NEWTON_ONE_SITE
#define RMX_BIG_NEWTON_ONE_SITE 1
#endif
#ifndef RMX_BIG_NEWTON_ROT
#define RMX_BIG_NEWTON_ROT 1             // the rotated Newton loop (0: the reference's order of evaluations; build variants)
#endif
#ifndef RMX_BIG_PIVOT_INLINE
#define RMX_BIG_PIVOT_INLINE 0
#endif
#ifndef RMX_BIG_HESS_MFMA
#define RMX_BIG_HESS_MFMA 1             // the Hessian's two products on the matrix cores (0: the column loop; build variants)
#endif

// The per-node workspace lives in LDS (the dynamic array `dyn`, rows of per-node data are [component][n]); only H goes to global memory
// when it does not fit next to it.  Offsets in doubles; region X is reused: E, V (path products / sums) -> the in-place suffix scan -> H.
struct BigWs {
    double* H;        // [nr][nr] column-major, global memory (used when !HL)
    int ns;           // stride of the per-node rows = number of nodes
    int oE[2];        // [12][ns] x 2  world transforms (R row-major 9, p 3), double-buffered for the pointer jumping
    int oV[2];        // [6][ns] x 2   path sums (phi, beta)
    int oS;           // [28][ns]      body terms -> suffix sums (in place)
    int ocu;          // [6][ns]       y - z          (column i as seen from its strict ancestors)
    int ocl;          // [12][ns]      m1, m2w, sw    (column i as seen from its strict descendants); [18][ns] with ground contact: + m2v, sv
    int ncl;          // 12 or 18
};
__host__ __device__ constexpr size_t big_ws_doubles_n(const int nr) { return (size_t)nr * nr; }
// doubles of LDS: region X = max(E + V, scan, H if it is kept in LDS) + cu, cl
constexpr int LU_NB = 32;             // panel width of the blocked LU (H in HBM)
__host__ __device__ constexpr size_t big_lds_doubles(const int n, const int nr, const bool hl, const bool ct) {
    // region X: E + V (36 n) | the scan (28 n) | H (hl) or the LU panel [LU_NB][nr] + its U12 block [LU_NB][BT] (!hl); then cu, cl
    // (H in LDS is formed while cu, cl are read: behind them; the LU panel is only used after H is complete: over everything)
    const size_t x0 = (size_t)36 * n, cc = (size_t)(ct ? 24 : 18) * n;
    if (hl) return ((size_t)nr * nr > x0 ? (size_t)nr * nr : x0) + cc;
    const size_t lu = (size_t)LU_NB * nr + (size_t)LU_NB * BT;
    return lu > x0 + cc ? lu : x0 + cc;
}
template <bool HL>
__device__ __forceinline__ BigWs big_ws(double* base, const int n, const int nr, const bool ct = false) {
    BigWs w;
    w.H = base;
    w.ns = n;
    w.oE[0] = 0; w.oE[1] = 12 * n;
    w.oV[0] = 24 * n; w.oV[1] = 30 * n;
    w.oS = 0;
    const int xsz = (HL && nr * nr > 36 * n) ? nr * nr : 36 * n;
    w.ocu = xsz;
    w.ocl = xsz + 6 * n;
    w.ncl = ct ? 18 : 12;
    return w;
}

// H in LDS (HL): up to ~136 reduced DOFs the nr x nr matrix fits the CU's 160 KB next to the small static arrays, and one pivot
// step of the LU is a few LDS round trips instead of global ones (the solve is latency-bound: thread = row, a barrier per pivot).
// The accesses name the LDS array itself: no generic pointer into LDS is ever formed (see block_sum).
extern __shared__ __attribute__((aligned(16))) double dyn[];
template <bool HL>
__device__ __forceinline__ double hget(const double* __restrict__ Hg, const size_t i) {
    if constexpr (HL) return dyn[i];
    else return Hg[i];
}
template <bool HL>
__device__ __forceinline__ void hput(double* __restrict__ Hg, const size_t i, const double v) {
    if constexpr (HL) dyn[i] = v;
    else Hg[i] = v;
}

// sum over the workgroup, identical in every thread (fixed tree order).  (The LDS arrays are function-local statics, not pointer
// arguments: a generic pointer into LDS handed to an out-of-line device function trips a compiler bug on gfx950 - an illegal
// V_CMP_NE_U32 against src_shared_base.)
__device__ __forceinline__ double block_sum(double v, const int t) {
    __shared__ double sred[BT / 64];
    const double ws = wave_sum(v);          // identical in every lane of the wavefront (rmx_device.h)
    if ((t & 63) == 0) sred[t >> 6] = ws;
    __syncthreads();
    double r = sred[0];
#pragma unroll
    for (int k = 1; k < BT / 64; ++k) r += sred[k];
    __syncthreads();
    return r;
}
// (value, index) of the largest v over the WAVEFRONT, the lowest index among equals (dgetf2's first maximum), identical in every lane;
// v >= -1, a NaN never wins.  Butterfly inside the 16-lane rows on DPP moves, the four row winners through v_readlane: ~60 VALU
// instructions.  (The __shfl_xor form - six dependent ds_bpermute round trips of three values - was ~1.2 k ticks of every pivot's
// serial path.)
__device__ __forceinline__ void wave_argmax(double& v, int& idx) {
    auto take = [&](const double ov, const int oi) {
        const bool w = ov > v || (ov == v && oi < idx);
        v = w ? ov : v;
        idx = w ? oi : idx;
    };
    take(dpp_d<DPP_XOR1>(v), dpp_i<DPP_XOR1>(idx));
    take(dpp_d<DPP_XOR2>(v), dpp_i<DPP_XOR2>(idx));
    take(dpp_d<DPP_HALF_MIRROR>(v), dpp_i<DPP_HALF_MIRROR>(idx));
    take(dpp_d<DPP_MIRROR>(v), dpp_i<DPP_MIRROR>(idx));
    const double v0 = readlane_d(v, 0), v1 = readlane_d(v, 16), v2 = readlane_d(v, 32), v3 = readlane_d(v, 48);
    const int i0 = __builtin_amdgcn_readlane(idx, 0), i1 = __builtin_amdgcn_readlane(idx, 16), i2 = __builtin_amdgcn_readlane(idx, 32),
              i3 = __builtin_amdgcn_readlane(idx, 48);
    v = v0; idx = i0;
    take(v1, i1);
    take(v2, i2);
    take(v3, i3);
}
// index of the largest v over the workgroup, the lowest index among equals; v >= -1, NaN never wins
__device__ __forceinline__ int block_argmax(double v, int idx, const int t) {
    __shared__ double sv[BT / 64];
    __shared__ int si[BT / 64];
    wave_argmax(v, idx);
    if ((t & 63) == 0) {
        sv[t >> 6] = v;
        si[t >> 6] = idx;
    }
    __syncthreads();
    double bv = sv[0];
    int bi = si[0];
#pragma unroll
    for (int k = 1; k < BT / 64; ++k)
        if (sv[k] > bv || (sv[k] == bv && si[k] < bi)) {
            bv = sv[k];
            bi = si[k];
        }
    __syncthreads();
    return bi;
}
__device__ __forceinline__ int block_all(const bool v, const int t) {
    __shared__ int sflag;
    if (t == 0) sflag = 1;
    __syncthreads();
    if (!v) sflag = 0;        // benign race: every writer stores 0
    __syncthreads();
    const int r = sflag;
    __syncthreads();
    return r;
}

__device__ __forceinline__ int block_any(const bool v, const int t) {
    __shared__ int sany;
    if (t == 0) sany = 0;
    __syncthreads();
    if (v) sany = 1;          // benign race: every writer stores 1
    __syncthreads();
    const int r = sany;
    __syncthreads();
    return r;
}

// This node's constants: the model's arrays, or - for a node of a JointSpherical group - the variant of the group's current chart
struct NodeConsts {
    const double* K;   int ks;     // K[r * ks], r = 0..35
    const double* sb;  int ss;     // sb[r * ss], r = 0..5
};
__device__ __forceinline__ NodeConsts node_consts(const DevModel& M, const int t, const int* chart) {
    NodeConsts c;
    c.K = M.K + t; c.ks = M.stride;
    c.sb = M.sb + t; c.ss = M.stride;
    for (int g = 0; g < M.nsph; ++g) {
        const int k = t - M.sph_first[g];
        if (k >= 0 && k < 3) {
            int a1, a2, a3;
            chart_axes(chart[g], a1, a2, a3);
            const int a = k == 0 ? a1 : (k == 1 ? a2 : a3);
            c.K = M.sphV + ((size_t)(g * 3 + k) * 3 + a) * SPH_ROWS; c.ks = 1;
            c.sb = c.K + 36; c.ss = 1;
        }
    }
    return c;
}

struct BigOut {
    double g, eT, eV;
};

// evalBDF1 / computeValues for the generic implicit residual (see eval_front_e2 / eval_hess in rmx_device.h for the wavefront form and
// oracle/redmax_tensorfree.c tf_eval for the node-by-node restatement this follows).  Thread t = node t; q, qd, v are this node's.
struct BigNoGate {
    __device__ __forceinline__ bool operator()(double) const { return false; }
};
// gate(g), workgroup-uniform, is asked once the residual is complete: true ends the evaluation there, before the Hessian's share of
// the work (big_newton_inl: a line-search trial that is rejected, or accepted and converged, needs no H).
template <bool WANT_H, bool HL = false, bool CT = false, class Gate = BigNoGate>
__device__ __forceinline__ void big_eval(const DevModel& M, const BigWs& w, const NodeConsts& nc, const int t, const double q, const double qd,
                         const double v, const double eta, BigOut& out, Gate&& gate = Gate()) {
    const int n = M.n, NS = M.stride, LS = w.ns;      // NS: stride of the model's constant tables, LS: of the LDS rows
    const bool act = t < n;
    const int tj = act ? t : 0;
    const double e2 = eta * eta;
    const int type = act ? M.type[tj] : 0;
    const bool dof = type != 0;
    // ---- joint transform T_j(q) = K0 + u K1 + w K2
    double u = 0.0, ww = 0.0;
    if (type == 1) sincos(q, &u, &ww);
    else if (type == 2) u = q;
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, p[3] = {0, 0, 0};
    if (act) {
#pragma unroll
        for (int c = 0; c < 9; ++c) R[c] = nc.K[c * nc.ks] + u * nc.K[(12 + c) * nc.ks] + ww * nc.K[(24 + c) * nc.ks];
#pragma unroll
        for (int c = 0; c < 3; ++c) p[c] = nc.K[(9 + c) * nc.ks] + u * nc.K[(21 + c) * nc.ks] + ww * nc.K[(33 + c) * nc.ks];
    }
    // ---- world transforms: pointer jumping, E_w,j = E_w,anc T_j...
    int cur = 0;
    if (act) {
#pragma unroll
        for (int c = 0; c < 9; ++c) dyn[w.oE[0] + c * LS + t] = R[c];
#pragma unroll
        for (int c = 0; c < 3; ++c) dyn[w.oE[0] + (9 + c) * LS + t] = p[c];
    }
    __syncthreads();
    for (int r = 0; r < M.rounds; ++r) {
        const int a = act ? M.anc[r * NS + tj] : -1;
        if (a >= 0) {
            double Ra[9], pa[3];
#pragma unroll
            for (int c = 0; c < 9; ++c) Ra[c] = dyn[w.oE[cur] + c * LS + a];
#pragma unroll
            for (int c = 0; c < 3; ++c) pa[c] = dyn[w.oE[cur] + (9 + c) * LS + a];
            double Rn[9], pn[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
#pragma unroll
                for (int k = 0; k < 3; ++k) Rn[3 * i + k] = Ra[3 * i] * R[k] + Ra[3 * i + 1] * R[3 + k] + Ra[3 * i + 2] * R[6 + k];
                pn[i] = Ra[3 * i] * p[0] + Ra[3 * i + 1] * p[1] + Ra[3 * i + 2] * p[2] + pa[i];
            }
#pragma unroll
            for (int c = 0; c < 9; ++c) R[c] = Rn[c];
#pragma unroll
            for (int c = 0; c < 3; ++c) p[c] = pn[c];
        }
        if (act) {
#pragma unroll
            for (int c = 0; c < 9; ++c) dyn[w.oE[cur ^ 1] + c * LS + t] = R[c];
#pragma unroll
            for (int c = 0; c < 3; ++c) dyn[w.oE[cur ^ 1] + (9 + c) * LS + t] = p[c];
        }
        cur ^= 1;
        __syncthreads();
    }
    // ---- world-frame joint screw
    double sw[3] = {0, 0, 0}, sv[3] = {0, 0, 0}, t3[3];
    if (act) {
        const double sbw[3] = {nc.sb[0], nc.sb[nc.ss], nc.sb[2 * nc.ss]}, sbv[3] = {nc.sb[3 * nc.ss], nc.sb[4 * nc.ss], nc.sb[5 * nc.ss]};
        mat3v(R, sbw, sw);
        mat3v(R, sbv, sv);
        cross3(p, sw, t3);
#pragma unroll
        for (int c = 0; c < 3; ++c) sv[c] += t3[c];
    }
    // path sum of a 6-vector (own value in x6, result in x6)
    auto path_sum6 = [&](double (&x6)[6]) {
        int cv = 0;
        __syncthreads();          // the previous pass (or the transforms) may still be reading this region
        if (act) {
#pragma unroll
            for (int c = 0; c < 6; ++c) dyn[w.oV[0] + c * LS + t] = x6[c];
        }
        __syncthreads();
        for (int r = 0; r < M.rounds; ++r) {
            const int a = act ? M.anc[r * NS + tj] : -1;
            if (a >= 0) {
#pragma unroll
                for (int c = 0; c < 6; ++c) x6[c] += dyn[w.oV[cv] + c * LS + a];
            }
            if (act) {
#pragma unroll
                for (int c = 0; c < 6; ++c) dyn[w.oV[cv ^ 1] + c * LS + t] = x6[c];
            }
            cv ^= 1;
            __syncthreads();
        }
    };
    double ph[6], be[6];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        ph[c] = sw[c] * qd;
        ph[3 + c] = sv[c] * qd;
    }
    path_sum6(ph);
    const double* phw = ph;
    const double* phv = ph + 3;
    double xiw[3], xiv[3];
    cross3(phw, sw, xiw);
    cross3(phv, sw, xiv);
    cross3(phw, sv, t3);
#pragma unroll
    for (int c = 0; c < 3; ++c) xiv[c] += t3[c];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        be[c] = sw[c] * v + e2 * qd * xiw[c];
        be[3 + c] = sv[c] * v + e2 * qd * xiv[c];
    }
    path_sum6(be);
    const double* bw = be;
    const double* bv = be + 3;
    // ---- body terms (Body.computeMassGrav): world-frame inertia about the origin, momentum, wrench, B block
    const double I1 = act ? M.I4[0 * NS + tj] : 0.0, I2 = act ? M.I4[1 * NS + tj] : 0.0, I3 = act ? M.I4[2 * NS + tj] : 0.0, ms = act ? M.I4[3 * NS + tj] : 0.0;
    double mc[3], Ib[6];
#pragma unroll
    for (int c = 0; c < 3; ++c) mc[c] = ms * p[c];
    {
        const double cc = dot3(p, p);
        Ib[0] = I1 * R[0] * R[0] + I2 * R[1] * R[1] + I3 * R[2] * R[2] + ms * (cc - p[0] * p[0]);
        Ib[1] = I1 * R[0] * R[3] + I2 * R[1] * R[4] + I3 * R[2] * R[5] - ms * p[0] * p[1];
        Ib[2] = I1 * R[0] * R[6] + I2 * R[1] * R[7] + I3 * R[2] * R[8] - ms * p[0] * p[2];
        Ib[3] = I1 * R[3] * R[3] + I2 * R[4] * R[4] + I3 * R[5] * R[5] + ms * (cc - p[1] * p[1]);
        Ib[4] = I1 * R[3] * R[6] + I2 * R[4] * R[7] + I3 * R[5] * R[8] - ms * p[1] * p[2];
        Ib[5] = I1 * R[6] * R[6] + I2 * R[7] * R[7] + I3 * R[8] * R[8] + ms * (cc - p[2] * p[2]);
    }
    double ht[3], hf[3], bt[3], bf[3], a3[3], b3[3], c3[3], fgt[3];
    sym3v(Ib, phw, ht);
    cross3(mc, phv, t3);
#pragma unroll
    for (int c = 0; c < 3; ++c) ht[c] += t3[c];
    cross3(mc, phw, t3);
#pragma unroll
    for (int c = 0; c < 3; ++c) hf[c] = ms * phv[c] - t3[c];
    sym3v(Ib, bw, bt);
    cross3(mc, bv, t3);
#pragma unroll
    for (int c = 0; c < 3; ++c) bt[c] += t3[c];
    cross3(mc, bw, t3);
#pragma unroll
    for (int c = 0; c < 3; ++c) bf[c] = ms * bv[c] - t3[c];
    const double gv[3] = {M.grav[0], M.grav[1], M.grav[2]};
    cross3(phw, ht, a3);
    cross3(phv, hf, b3);
    cross3(phw, hf, c3);
    cross3(mc, gv, fgt);
    double S[28];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        S[c] = bt[c] - e2 * (-a3[c] - b3[c] + fgt[c]);
        S[3 + c] = bf[c] - e2 * (-c3[c] + ms * gv[c]);
    }
    S[6] = ms;
#pragma unroll
    for (int c = 0; c < 3; ++c) S[7 + c] = mc[c];
#pragma unroll
    for (int c = 0; c < 6; ++c) S[10 + c] = Ib[c];
    {
        const double Ibf[9] = {Ib[0], Ib[1], Ib[2], Ib[1], Ib[3], Ib[4], Ib[2], Ib[4], Ib[5]};
        const double Om[9] = {0.0, -phw[2], phw[1], phw[2], 0.0, -phw[0], -phw[1], phw[0], 0.0};
        const double Vx[9] = {0.0, -phv[2], phv[1], phv[2], 0.0, -phv[0], -phv[1], phv[0], 0.0};
        const double Mc[9] = {0.0, -mc[2], mc[1], mc[2], 0.0, -mc[0], -mc[1], mc[0], 0.0};
        const double Ht[9] = {0.0, -ht[2], ht[1], ht[2], 0.0, -ht[0], -ht[1], ht[0], 0.0};
        double X[9];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                double tt = 0.0;
#pragma unroll
                for (int l = 0; l < 3; ++l) tt += Ibf[3 * i + l] * Om[3 * l + k] + Mc[3 * i + l] * Vx[3 * l + k];
                X[3 * i + k] = tt;
            }
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int k = 0; k < 3; ++k) S[16 + 3 * i + k] = X[3 * i + k] + X[3 * k + i] + Ht[3 * i + k];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) S[25 + c] = hf[c];
    // ---- ground contact (ForceGroundCuboid.m:54-153; contact_body in rmx_device.h, the world-frame form of DESIGN.md section 3): the
    // wrench of this body's penetrating corners goes into w_b; its K / D blocks are formed in the Hessian part below
    GroundC G;
    bool con = false, touched = false;
    double sd[3] = {0.0, 0.0, 0.0}, eVc = 0.0;
    if constexpr (CT) {
        con = act && M.con[tj] != 0.0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            sd[c] = M.con[(1 + c) * NS + tj];
            G.n[c] = M.con[(4 + c) * NS + tj];
            G.gx[c] = M.con[(7 + c) * NS + tj];
        }
        G.kn = M.con[10 * NS + tj];
        G.kt = M.con[11 * NS + tj];
        G.mu = M.con[12 * NS + tj];
        G.kdc = M.con[13 * NS + tj];
        double Fc[6], k1[36];
        const bool tw = contact_body<0>(G, con, sd, R, p, phw, phv, Fc, k1, eVc);      // wave-uniform: some corner of this wave's bodies
#pragma unroll
        for (int c = 0; c < 6; ++c) S[c] -= e2 * Fc[c];
        if (WANT_H) touched = block_any(tw, t) != 0;
    }
    // energies (Body.computeEnergies, Joint.computeEnergies)
    const double stiff = act ? M.prm[1 * NS + tj] : 0.0, damp = act ? M.prm[2 * NS + tj] : 0.0;
    const double tau = act ? M.prm[0 * NS + tj] : 0.0, qRest = act ? M.prm[3 * NS + tj] : 0.0;
    const double qLimL = act ? M.prm[4 * NS + tj] : 0.0, qLimU = act ? M.prm[5 * NS + tj] : 0.0;
    const double qLimK = act ? M.prm[6 * NS + tj] : 0.0, qLimD = act ? M.prm[7 * NS + tj] : 0.0;
    const double hitL = (dof && q < qLimL) ? 1.0 : 0.0, hitU = (dof && q > qLimU) ? 1.0 : 0.0;
    {
        out.eT = act ? 0.5 * (dot3(phw, ht) + dot3(phv, hf)) : 0.0;
        double eV = act ? -dot3(gv, mc) : 0.0;
        if (dof) {
            const double dq = q - qRest;
            const double dqL = hitL * (qLimL - q), dqU = hitU * (qLimU - q);
            eV += 0.5 * stiff * (dq * dq) + 0.5 * qLimK * (dqL * dqL + dqU * dqU);
        }
        out.eV = eV + eVc;
    }
    // ---- subtree sums: suffix scan over the depth-first order (Hillis-Steele, in place: read, barrier, write), subtree(j) =
    // suffix(j) - suffix(end_j).  The region was E / V: every path product and sum is in registers by now.
    auto subtree_sum = [&](double* X, auto ncTag) {
        constexpr int NC = decltype(ncTag)::value;
        __syncthreads();
        if (act) {
#pragma unroll
            for (int c = 0; c < NC; ++c) dyn[w.oS + c * LS + t] = X[c];
        }
        __syncthreads();
        for (int d = 1; d < n; d <<= 1) {
            double add[NC];
            const bool on = act && t + d < n;
#pragma unroll
            for (int c = 0; c < NC; ++c) add[c] = on ? dyn[w.oS + c * LS + t + d] : 0.0;
            __syncthreads();
            if (act) {
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    X[c] += add[c];
                    dyn[w.oS + c * LS + t] = X[c];
                }
            }
            __syncthreads();
        }
        const int en = act ? M.end[tj] : n;
        if (en < n) {
#pragma unroll
            for (int c = 0; c < NC; ++c) X[c] -= dyn[w.oS + c * LS + en];
        }
        __syncthreads();
    };
    subtree_sum(S, std::integral_constant<int, 28>{});
    // ---- residual  g_j = s_j . W_j - eta^2 fr_j
    const double fr = tau + stiff * (qRest - q) - damp * qd + hitL * (qLimK * (qLimL - q) - qLimD * qd) + hitU * (qLimK * (qLimU - q) - qLimD * qd);
    out.g = dof ? (dot3(sw, &S[0]) + dot3(sv, &S[3]) - e2 * fr) : 0.0;
    if (!WANT_H) return;
    if (gate(out.g)) return;
    // ---- Hessian vectors of this node (eval_hess in rmx_device.h)
    const double kd = stiff + (hitL + hitU) * qLimK, dd = damp + (hitL + hitU) * qLimD;
    const double* Wt = &S[0];
    const double* Wf = &S[3];
    const double mS = S[6];
    const double* mcS = &S[7];
    const double* IbS = &S[10];
    const double* TL = &S[16];
    const double* hfS = &S[25];
    double zw[3], zv[3], m1w[3], m1v[3], m2w[3];
    cross3(bw, sw, zw);
    cross3(bv, sw, zv);
    cross3(bw, sv, t3);
#pragma unroll
    for (int c = 0; c < 3; ++c) zv[c] += t3[c];
    cross3(phw, xiw, a3);
    cross3(phv, xiw, b3);
    cross3(phw, xiv, t3);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        zw[c] += e2 * a3[c];
        zv[c] += e2 * (b3[c] + t3[c]);
        m1w[c] = sw[c] + 2.0 * eta * xiw[c] + zw[c];
        m1v[c] = sv[c] + 2.0 * eta * xiv[c] + zv[c];
        m2w[c] = eta * sw[c] + e2 * xiw[c];
    }
    // ground contact: K / D blocks of this body at this state, summed over the subtree (36 + 21 numbers through the scan above) and
    // folded into cxy = Dxc m2 + eta^2 Kxc s, cxr2 = Dxc' s, cxr3 = eta^2 Kxc' s (eval_hess in rmx_device.h); skipped while no
    // corner of the tree penetrates (workgroup-uniform)
    double cxy[6] = {0, 0, 0, 0, 0, 0}, cxr2[6] = {0, 0, 0, 0, 0, 0}, cxr3[6] = {0, 0, 0, 0, 0, 0};
    if (CT && touched) {
        const double s6[6] = {sw[0], sw[1], sw[2], sv[0], sv[1], sv[2]};
        double m26[6], Fc[6], ev2;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            m26[c] = eta * sw[c] + e2 * xiw[c];
            m26[3 + c] = eta * sv[c] + e2 * xiv[c];
        }
        {
            double Kx[36];
            contact_body<1>(G, con, sd, R, p, phw, phv, Fc, Kx, ev2);
            subtree_sum(&Kx[0], std::integral_constant<int, 28>{});
            subtree_sum(&Kx[28], std::integral_constant<int, 8>{});
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                double ay = 0.0, a3k = 0.0;
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    ay += Kx[6 * r + c] * s6[c];
                    a3k += Kx[6 * c + r] * s6[c];
                }
                cxy[r] = e2 * ay;
                cxr3[r] = e2 * a3k;
            }
        }
        {
            double Dx[36];
            contact_body<2>(G, con, sd, R, p, phw, phv, Fc, Dx, ev2);
            subtree_sum(&Dx[0], std::integral_constant<int, 21>{});
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                double ay = 0.0, a2 = 0.0;
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    const double drc = Dx[r <= c ? sym21(r, c) : sym21(c, r)];
                    ay += drc * m26[c];
                    a2 += drc * s6[c];
                }
                cxy[r] += ay;
                cxr2[r] = a2;
            }
        }
    }
    double yt[3], yf[3], gxs[3], kt[3];
    sym3v(IbS, m1w, yt);
    cross3(mcS, m1v, t3);
#pragma unroll
    for (int c = 0; c < 3; ++c) yt[c] += t3[c];
    cross3(mcS, m1w, t3);
#pragma unroll
    for (int c = 0; c < 3; ++c) yf[c] = mS * m1v[c] - t3[c];
    mat3v(TL, m2w, a3);
    cross3(hfS, m2w, b3);
    cross3(gv, sw, gxs);
    cross3(mcS, gxs, kt);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        yt[c] -= a3[c] + e2 * kt[c] + cxy[c];
        yf[c] -= 2.0 * b3[c] + e2 * mS * gxs[c] + cxy[3 + c];
    }
    double zt[3], zf[3];
    cross3(sw, Wt, a3);
    cross3(sv, Wf, b3);
    cross3(sw, Wf, zf);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        zt[c] = -a3[c] - b3[c];
        zf[c] = -zf[c];
    }
    const double Hdiag = dof ? (dot3(sw, yt) + dot3(sv, yf) + eta * dd + e2 * kd) : 1.0;
    double rl[18];          // r1, -r2w, -r3w [, -cxr2(4:6), -cxr3(4:6): the ground contact's rows against m2v, sv]
    sym3v(IbS, sw, rl);
    cross3(mcS, sv, t3);
#pragma unroll
    for (int c = 0; c < 3; ++c) rl[c] += t3[c];
    cross3(mcS, sw, t3);
#pragma unroll
    for (int c = 0; c < 3; ++c) rl[3 + c] = mS * sv[c] - t3[c];
    cross3(hfS, sv, b3);
#pragma unroll
    for (int c = 0; c < 3; ++c) rl[6 + c] = -(TL[c] * sw[0] + TL[3 + c] * sw[1] + TL[6 + c] * sw[2] - 2.0 * b3[c]) - cxr2[c];
    cross3(gv, t3, a3);
    cross3(gv, sv, b3);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        rl[9 + c] = -e2 * (a3[c] - mS * b3[c]) - cxr3[c];
        rl[12 + c] = -cxr2[3 + c];
        rl[15 + c] = -cxr3[3 + c];
    }
    const int ncl = (CT && touched) ? 18 : 12;  // workgroup-uniform (CT implies the 18-row layout: w.ncl == 18)
    __shared__ short s_idx[BT], s_end[BT];      // reduced index and subtree end of every node: what the column loop below asks of node i
    s_idx[t] = act ? (short)M.idx[tj] : (short)-1;
    s_end[t] = act ? (short)M.end[tj] : (short)0;
    if (act) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            dyn[w.ocu + c * LS + t] = yt[c] - zt[c];
            dyn[w.ocu + (3 + c) * LS + t] = yf[c] - zf[c];
            dyn[w.ocl + c * LS + t] = m1w[c];
            dyn[w.ocl + (3 + c) * LS + t] = m1v[c];
            dyn[w.ocl + (6 + c) * LS + t] = m2w[c];
            dyn[w.ocl + (9 + c) * LS + t] = sw[c];
            if (CT && touched) {
                dyn[w.ocl + (12 + c) * LS + t] = eta * sv[c] + e2 * xiv[c];
                dyn[w.ocl + (15 + c) * LS + t] = sv[c];
            }
        }
    }
    __syncthreads();
    // ---- row of this node: H(a, i) = s_a . cu_i for a a strict ancestor of i, rl_a . cl_i for a strict descendant, Hdiag on the
    // diagonal, 0 otherwise.  Depth-first order: a is a strict ancestor of i iff a < i < end_a.  Both products are formed for every
    // column and the relation selects: uniform control flow, so four columns' worth of (broadcast) LDS reads are in flight at a time.
    // The branchy form - pick the product by relation, three divergent bodies per column behind two scalar loads of idx / end - cost
    // ~1.7 k ticks per column (in-kernel timers, tools/big_profile.py), a fifth of a Newton iteration at 128 DOFs.
    const int nr = M.nr;
    const int ka = act ? M.idx[tj] : -1;
    if constexpr (RMX_BIG_HESS_MFMA && HL) {
    // The two products ARE matrix products - H_upper = S (n x 6) CU (6 x n) above the diagonal, H_lower = RL (n x 12 | 18) CL below - and
    // run on the fp64 matrix cores, tile by tile (v_mfma_f64_16x16x4_f64, the transposed form of the LU's trailing update: a lane's four
    // results are four columns of one row, a store touches 16 consecutive rows).  A wavefront takes the four row blocks of its own 64
    // rows: the row-side operand (this node's s / rl, registers) reaches the lanes that need it through ds_bpermute, no staging; the
    // column-side operand is the [component][node] rows of cu / cl as they lie in LDS.  Tiles right of the diagonal block need the
    // ancestors' product only (2 MFMAs), left of it the descendants' (3, or 5 with ground contact), the diagonal block both.
    // (The column loop it replaces: 6 / 12 / 18 LDS broadcasts and as many FMAs per column and row.  Measured, ticks per Newton iteration:
    // 72 links 248 -> 238 k, 128 links 389 -> 377 k.  H in HBM keeps the column loop: a thread per row stores 64 consecutive rows of a
    // column at a time, the tiles 16, and at 256 links what the products gain the stores lose - 1 155 vs 1 178 k.)
    {
        typedef double v4d __attribute__((ext_vector_type(4)));
        const int lane = t & 63, jj = lane & 15, gg = lane >> 4, wv = t >> 6;
        const int NBK = (n + 15) >> 4;
        const int ea_t = act ? M.end[tj] : 0;
        double* __restrict__ Hw = w.H;
        auto pick = [&](const double x0, const double x1, const double x2, const double x3) {
            return gg == 0 ? x0 : (gg == 1 ? x1 : (gg == 2 ? x2 : x3));
        };
#pragma unroll 1
        for (int mbl = 0; mbl < 4; ++mbl) {
            const int mb = 4 * wv + mbl;
            if (mb >= NBK) break;                    // wavefront-uniform
            const int src = 16 * mbl + jj;           // the lane of this wavefront that holds row a
            const int a = 16 * mb + jj;
            const int ka_a = __shfl(ka, src, 64), ea_a = __shfl(ea_t, src, 64);
            const double hd_a = __shfl(Hdiag, src, 64);
            double su[6];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                su[c] = __shfl(sw[c], src, 64);
                su[3 + c] = __shfl(sv[c], src, 64);
            }
            const double bU0 = pick(su[0], su[1], su[2], su[3]), bU1 = pick(su[4], su[5], 0.0, 0.0);
            double bL[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < 3; ++kk)
                bL[kk] = pick(__shfl(rl[4 * kk], src, 64), __shfl(rl[4 * kk + 1], src, 64), __shfl(rl[4 * kk + 2], src, 64), __shfl(rl[4 * kk + 3], src, 64));
            if (CT && ncl == 18) {
                bL[3] = pick(__shfl(rl[12], src, 64), __shfl(rl[13], src, 64), __shfl(rl[14], src, 64), __shfl(rl[15], src, 64));
                bL[4] = pick(__shfl(rl[16], src, 64), __shfl(rl[17], src, 64), 0.0, 0.0);
            }
            for (int nbk = 0; nbk < NBK; ++nbk) {
                const int ia = 16 * nbk + jj;        // this lane's column node in the column-side operand
                const bool iaon = ia < n;
                const int ic = iaon ? ia : 0;
                v4d hu = {0.0, 0.0, 0.0, 0.0}, hl = {0.0, 0.0, 0.0, 0.0};
                if (nbk >= mb) {                     // wavefront-uniform
                    const double a0 = iaon ? dyn[w.ocu + gg * LS + ic] : 0.0;
                    const double a1 = (iaon && gg < 2) ? dyn[w.ocu + (4 + (gg & 1)) * LS + ic] : 0.0;
                    hu = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, bU0, hu, 0, 0, 0);
                    hu = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, bU1, hu, 0, 0, 0);
                }
                if (nbk <= mb) {
#pragma unroll
                    for (int kk = 0; kk < 3; ++kk) {
                        const double al = iaon ? dyn[w.ocl + (4 * kk + gg) * LS + ic] : 0.0;
                        hl = __builtin_amdgcn_mfma_f64_16x16x4f64(al, bL[kk], hl, 0, 0, 0);
                    }
                    if (CT && ncl == 18) {
                        const double a3 = iaon ? dyn[w.ocl + (12 + gg) * LS + ic] : 0.0;
                        const double a4 = (iaon && gg < 2) ? dyn[w.ocl + (16 + (gg & 1)) * LS + ic] : 0.0;
                        hl = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, bL[3], hl, 0, 0, 0);
                        hl = __builtin_amdgcn_mfma_f64_16x16x4f64(a4, bL[4], hl, 0, 0, 0);
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i = 16 * nbk + 4 * q + gg;
                    if (i < n && a < n) {
                        const int ki = s_idx[i], ei = s_end[i];
                        const bool anc = a < i && i < ea_a, desc = i < a && a < ei;
                        const double h = (i == a) ? hd_a : (anc ? hu[q] : (desc ? hl[q] : 0.0));
                        if (ki >= 0 && ka_a >= 0) hput<HL>(Hw, (size_t)ki * nr + ka_a, h);
                    }
                }
            }
        }
    }
    } else {
    // A wavefront holds 64 consecutive rows: a column to the left of all of them can only be an ancestor's (the descendants' product
    // alone, 12 broadcast reads), one to the right of all of them only a descendant's (6); both products only for the wavefront's own 64
    // columns.  (Both for every column: 18 reads per column, the LDS pipe the bound of the loop - 25 .. 50 % more than needed.)
    if (ka >= 0) {
        const int ea = M.end[tj];
        double* __restrict__ Hw = w.H;
        auto column = [&](const int i, auto wantU, auto wantL) {
            const int ki = s_idx[i], ei = s_end[i];
            double hu = 0.0, hl = 0.0;
            if constexpr (decltype(wantU)::value) {
#pragma unroll
                for (int c = 0; c < 3; ++c) hu += sw[c] * dyn[w.ocu + c * LS + i] + sv[c] * dyn[w.ocu + (3 + c) * LS + i];
            }
            if constexpr (decltype(wantL)::value) {
#pragma unroll
                for (int c = 0; c < 12; ++c) hl += rl[c] * dyn[w.ocl + c * LS + i];
                if (CT && ncl == 18) {
#pragma unroll
                    for (int c = 12; c < 18; ++c) hl += rl[c] * dyn[w.ocl + c * LS + i];
                }
            }
            const bool anc = t < i && i < ea, desc = i < t && t < ei;
            const double h = (i == t) ? Hdiag : (anc ? hu : (desc ? hl : 0.0));
            if (ki >= 0) hput<HL>(Hw, (size_t)ki * nr + ka, h);
        };
        const int wlo = (t & ~63) < n ? (t & ~63) : n, whi = wlo + 64 < n ? wlo + 64 : n;     // wavefront-uniform
        int i = 0;
#pragma unroll 4
        for (; i < wlo; ++i) column(i, std::false_type{}, std::true_type{});
#pragma unroll 4
        for (; i < whi; ++i) column(i, std::true_type{}, std::true_type{});
#pragma unroll 4
        for (; i < n; ++i) column(i, std::true_type{}, std::false_type{});
    }
    }
    __syncthreads();
}

// dx = -H\g: LU with partial pivoting on the workspace copy of H (column-major nr x nr), thread = reduced row.  The permutation is
// implicit (rows are never moved): piv[r] = the step at which row r served as the pivot row, or -1.  First maximum wins (dgetf2).
// bneg: this node's -g (nodes without a DOF: ignored).  Returns this node's dx.
// The linear solvers' small arrays: ONE set for the three of them (a kernel holds the guarded diagonal solve and its pivoting fallback;
// function-local arrays would be allocated once per solver).  Named directly, never through a pointer (see block_sum).
__shared__ double lu_b[BT];          // right-hand side, reduced order
__shared__ double lu_xs[BT];         // reciprocals of the pivots
__shared__ int lu_piv[BT];           // pivot row of step k (pivoting solvers)

#ifdef RMX_BIG_PROFILE
__device__ unsigned long long g_prof[8];
#define PROF_T0() const unsigned long long p0_ = __builtin_amdgcn_s_memtime()
#define PROF_ADD(i) do { if (t == 0 && blockIdx.x == 0) g_prof[i] += __builtin_amdgcn_s_memtime() - p0_; } while (0)
#else
#define PROF_T0() do {} while (0)
#define PROF_ADD(i) do {} while (0)
#endif
// dx = -H\g with H in HBM (nr > ~128): right-looking BLOCKED LU with partial pivoting, thread = row, implicit permutation as in
// big_solve.  Unblocked, every pivot streams the whole trailing matrix through the CU (nr^3/3 x 16 B = 89 MB per solve at 256 DOFs:
// the launch was HBM-bound).  Here a panel of LU_NB columns is factored in LDS, the LU_NB pivot rows of every trailing column are
// forward-substituted (thread = column, U12 kept in LDS), and the trailing matrix is read and written ONCE per panel with the
// rank-LU_NB update a(r,c) -= sum_j L(r,j) U(j,c) (L(r,:) in registers, U from LDS).  Same pivots, same operations per entry as the
// unblocked loop up to the order of the subtractions.
template <bool HL>
__device__ __forceinline__ double big_solve_blocked(const DevModel& M, const BigWs& w, const int t, const int ka, const double g) {
    typedef double v4d __attribute__((ext_vector_type(4)));
    __shared__ int sused[BT];          // sused[r] != 0: row r has served as a pivot row (its multipliers are 0 from then on)
    __shared__ double spv[2][BT / 64];  // per-wavefront pivot candidates of the next panel column
    __shared__ int spi[2][BT / 64];
    // H in HBM: a 32-column panel is copied into LDS (dyn[j * nr + r]) and -U12 staged behind it.  H in LDS (HL): the panel is 16 columns
    // of H IN PLACE, -U12 is staged where the Hessian's column vectors were (cu, cl: dead once H is complete; 16 nr + nr <= 18 n doubles).
    constexpr int NB = HL ? 16 : LU_NB;
    const int nr = M.nr;
    double* __restrict__ H = w.H;
    const int oU = HL ? w.ocu : NB * nr;        // dyn[oU + j * sU + c]: -U12(j, c-th trailing column)
    const int sU = HL ? nr : BT;
    const int odummy = HL ? oU + NB * nr : NB * nr;     // where the panel update's accesses past the panel go (one column of nr doubles)
    auto HR = [&](const size_t i) -> double { if constexpr (HL) return dyn[i]; else return H[i]; };
    auto HW = [&](const size_t i, const double v) { if constexpr (HL) dyn[i] = v; else H[i] = v; };
    if (ka >= 0) lu_b[ka] = -g;
    sused[t] = 0;
    __syncthreads();
    const int r = t;
    const bool row = r < nr;
    int mystep = -1;
    for (int kb = 0; kb < nr; kb += NB) {
        const int nb = nr - kb < NB ? nr - kb : NB;
        const int pb = HL ? kb * nr : 0;            // panel column j, row r: dyn[pb + j * nr + r]
        PROF_T0();
        if constexpr (!HL) {
            if (row) {       // all NB loads in flight (a rolled loop is NB dependent trips to L2 / HBM)
                double pc[NB];
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    if (j < nb) pc[j] = H[(size_t)(kb + j) * nr + r];
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    if (j < nb) dyn[j * nr + r] = pc[j];
            }
            __syncthreads();
        }
        // The panel, pivot by pivot, in LDS; the multipliers stay in place.  ONE barrier per pivot: the search for pivot j + 1 rides inside
        // step j - its column is updated first, every wavefront reduces its rows' candidates on DPP butterflies while the remaining
        // panel columns are being updated, and the four per-wavefront winners cross in LDS (double-buffered by the parity of j) at the
        // barrier that ends the step.  (Search, two barriers, then update and a third barrier: 4.6 k ticks per pivot for ~600 ticks of
        // work, in-kernel timers.)  Same candidates, same comparisons (|a|, lowest row among equals), same multipliers.
        {
            double cv = (row && mystep < 0) ? fabs(dyn[pb + r]) : -1.0;
            int ci = r;
            wave_argmax(cv, ci);
            if ((t & 63) == 0) { spv[0][t >> 6] = cv; spi[0][t >> 6] = ci; }
        }
        __syncthreads();
        for (int j = 0; j < nb; ++j) {
            const int par = j & 1;
            double bv = spv[par][0];
            int pr = spi[par][0];
#pragma unroll
            for (int q = 1; q < BT / 64; ++q)
                if (spv[par][q] > bv || (spv[par][q] == bv && spi[par][q] < pr)) { bv = spv[par][q]; pr = spi[par][q]; }
            if (t == 0) lu_piv[kb + j] = pr;
            if (r == pr) {
                mystep = kb + j;
                sused[r] = 1;
            }
            const bool elim = row && mystep < 0;
            double l = 0.0, cv = -1.0;
            int ci = r;
            if (elim) {
                l = dyn[pb + j * nr + r] * recip(dyn[pb + j * nr + pr]);      // dgetf2 scales by the reciprocal of the pivot
                dyn[pb + j * nr + r] = l;
                if (j + 1 < nb) {                                  // the look-ahead column and this row's candidate for pivot j + 1
                    const double a1 = dyn[pb + (j + 1) * nr + r] - l * dyn[pb + (j + 1) * nr + pr];
                    dyn[pb + (j + 1) * nr + r] = a1;
                    cv = fabs(a1);
                }
            }
            wave_argmax(cv, ci);                                   // every lane takes part (finished rows and idle threads carry -1)
            if ((t & 63) == 0) { spv[par ^ 1][t >> 6] = cv; spi[par ^ 1][t >> 6] = ci; }
            if (elim) {
                // the remaining panel columns eight at a time, loads first (a rolled loop is one LDS round trip per column).  Columns past
                // the panel go to a dummy column - the first row of the U12 staging area behind the panel, dead until the panel is done -
                // instead of being guarded.
                for (int j0 = j + 2; j0 < nb; j0 += 8) {       // workgroup-uniform trip count
                    int cj[8];
                    double pv[8], av[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) cj[u] = j0 + u < nb ? pb + (j0 + u) * nr : odummy;
#pragma unroll
                    for (int u = 0; u < 8; ++u) pv[u] = dyn[cj[u] + pr];
#pragma unroll
                    for (int u = 0; u < 8; ++u) av[u] = dyn[cj[u] + r];
#pragma unroll
                    for (int u = 0; u < 8; ++u) dyn[cj[u] + r] = av[u] - l * pv[u];
                }
                lu_b[r] -= l * lu_b[pr];
            }
            __syncthreads();
        }
        if constexpr (!HL) {
            if (row) {   // U (pivot rows) for the back substitution; the multipliers of the other rows are never read from H again
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    if (j < nb) H[(size_t)(kb + j) * nr + r] = dyn[j * nr + r];
            }
        }
        PROF_ADD(3);
        // trailing columns (BT >= nr: one pass)
        const int c0 = kb + nb;
        const int ncol = nr - c0;
        if (ncol > 0) {
            PROF_T0();
            const int c = c0 + t;
            if (c < nr) {          // thread = column: U12(:, c) = L11^-1 A12(pivot rows, c)
                double u[NB];
                const size_t col = (size_t)c * nr;
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    double a = 0.0;
                    if (j < nb) {
                        const int pj = lu_piv[kb + j];
                        a = HR(col + pj);
#pragma unroll
                        for (int i = 0; i < j; ++i) a -= dyn[pb + i * nr + pj] * u[i];
                        HW(col + pj, a);
                    }
                    u[j] = a;
                    dyn[oU + j * sU + t] = -a;
                }
            }
            __syncthreads();
            PROF_ADD(4);
            // The rank-nb update A22 -= L21 U12 on the fp64 matrix cores: north_star's "MFMA when ndof is large enough to be a real
            // contraction" - up to 224 x 224 x 32 here, against K = 6 .. 12 in the 32-DOF Hessian.  Transposed form so that a lane's four
            // results are four COLUMNS of one row block and a load / store instruction touches 16 consecutive rows of 4 columns (H is
            // column-major): D'(col, row) = sum_k (-U12)(k, col) L21(row, k) + A22(row, col), v_mfma_f64_16x16x4_f64 with
            //   A operand  lane (jj, gg) -> (-U12)(k = 4 kk + gg, col = 16 nbk + jj)      from the LDS copy the pass above wrote
            //   B operand  lane (jj, gg) -> L21(row = 16 mb + jj, k = 4 kk + gg)          the panel in LDS; 0 for rows that have been pivots
            //   C / D      element q     -> A22(row = 16 mb + jj, col = 16 nbk + 4 q + gg)
            // Eight MFMAs per 16 x 16 tile (K = 32); a wavefront keeps the L fragments of a row block and walks the column blocks.
            // The sum over k runs in the same order as the scalar loop it replaces (one fused multiply-add per k).
            {
                const int wave = t >> 6, lane = t & 63, jj = lane & 15, gg = lane >> 4;
                const int MB = (nr + 15) >> 4, NBK = (ncol + 15) >> 4;
                for (int mb = wave; mb < MB; mb += BT / 64) {
                    const int arow = 16 * mb + jj;
                    const bool aon = arow < nr && sused[arow < nr ? arow : 0] == 0;
                    double lf[NB / 4];
#pragma unroll
                    for (int kk = 0; kk < NB / 4; ++kk) {
                        const int k = 4 * kk + gg;
                        lf[kk] = (aon && k < nb) ? dyn[pb + k * nr + arow] : 0.0;
                    }
                    v4d cn;                   // C of the tile about to be worked on, fetched one tile ahead
                    size_t adn[4];
                    bool okn[4];
                    auto fetch = [&](const int nbk) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int cc = 16 * nbk + 4 * q + gg;
                            okn[q] = cc < ncol && arow < nr;
                            adn[q] = (size_t)(c0 + (cc < ncol ? cc : 0)) * nr + (arow < nr ? arow : 0);
                            cn[q] = okn[q] ? HR(adn[q]) : 0.0;
                        }
                    };
                    fetch(0);
                    for (int nbk = 0; nbk < NBK; ++nbk) {
                        const int cj = 16 * nbk + jj;                    // this lane's column of the A operand
                        v4d acc = cn;
                        size_t ad[4];
                        bool ok[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            ad[q] = adn[q];
                            ok[q] = okn[q];
                        }
                        double uf[NB / 4];
#pragma unroll
                        for (int kk = 0; kk < NB / 4; ++kk) {
                            const int k = 4 * kk + gg;
                            uf[kk] = (cj < ncol && k < nb) ? dyn[oU + k * sU + cj] : 0.0;
                        }
                        if (nbk + 1 < NBK) fetch(nbk + 1);               // in flight underneath the eight MFMAs
#pragma unroll
                        for (int kk = 0; kk < NB / 4; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(uf[kk], lf[kk], acc, 0, 0, 0);
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (ok[q]) HW(ad[q], acc[q]);
                    }
                }
            }
            __syncthreads();
            PROF_ADD(5);
        }
    }
    // back substitution on the implicitly permuted upper triangle.  x_k = b(p_k) / U(p_k, k) is formed by EVERY thread from two broadcast
    // reads (the reciprocals of the pivots come from one parallel pass), so a step is one barrier instead of two and no thread waits for
    // the pivot row's owner; the column entries U(r, k) of eight steps are fetched together (each step used to wait for its own trip to
    // L2 / HBM).
    PROF_T0();
    if (row) lu_xs[r] = recip(HR((size_t)r * nr + lu_piv[r]));         // 1 / U(p_k, k), k = r
    __syncthreads();
    double dxr = 0.0;
    for (int k0 = nr - 1; k0 >= 0; k0 -= 8) {
        double u[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) u[i] = (row && k0 - i >= 0) ? HR((size_t)(k0 - i) * nr + r) : 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = k0 - i;
            if (k >= 0) {        // workgroup-uniform
                const double xk = lu_b[lu_piv[k]] * lu_xs[k];
                if (k == ka) dxr = xk;
                if (row && mystep < k) lu_b[r] -= u[i] * xk;
                __syncthreads();
            }
        }
    }
    PROF_ADD(6);
    return dxr;
}

// dx = -H\g WITHOUT the pivot search: right-looking blocked LU on the diagonal under the growth guard of the one-wavefront kernels
// (rmx_device.h lu_solve_neg_diag: every multiplier of the symmetrically equilibrated matrix below LU_GROWTH_MAX - H = M - eta D -
// eta^2 K is a modest perturbation of the SPD mass matrix, where that always holds; BigGrowGuard above for the sign).  ok = false: the
// caller re-assembles H and solves with partial pivoting (big_solve), so mldivide's semantics are kept and paid for only on demand.
// What the search cost: with a row permutation every pivot is a workgroup-wide argmax plus a barrier, ~3 k ticks of serial latency for
// ~600 ticks of work, 1.17 M of the 2.4 M ticks of a Newton iteration at 256 DOFs (in-kernel timers, profiles/r04i_big_profile.txt).
// Without it a panel is THREE barriers:
//   A  the wavefront that owns rows kb .. kb + NB - 1 (NB | 64: one wavefront) factors the diagonal block in registers - thread = row,
//      the pivot row broadcast with v_readlane - and eliminates the same columns from the other rows it owns on the same instructions;
//      the right-hand side rides along.  The other wavefronts have their panel rows and U12 columns in flight from HBM meanwhile.
//   B  thread = row r >= c0 of the other wavefronts: the same elimination with the rows of U11 broadcast from LDS;
//      thread = column c >= c0: U12(:, c) = L11^-1 A12(:, c), -U12 staged in LDS for C
//   C  A22 -= L21 U12 on the fp64 matrix cores, rows and columns >= c0 only (the permuted form had to sweep every row: 1.6 x the
//      traffic of a launch that is bound by it at 256 DOFs)
// Same elimination order per entry as big_solve_blocked on a matrix whose pivots are the diagonal.
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// The guard of big_solve_diag.  Unlike the one-wavefront solves it takes NEGATIVE pivots: on long chains H = M - eta D - eta^2 K is
// indefinite in a good part of the hard Newton iterations (every trip of the positive-pivot guard that tools/big_profile.py printed was a
// negative pivot, none a multiplier: 41 % of the rollouts of the 256-link bench tripped at least once), and there the fallback costs a
// re-assembly plus the pivoting solve, three times the guarded one.  What elimination on the diagonal needs for stability is bounded
// multipliers, not a sign: every multiplier of the symmetrically equilibrated matrix must satisfy l'^2 = |l a| / d_r <= LU_GROWTH_MAX^2
// (threshold pivoting with tau = 1 / 8 accepts the diagonal), every pivot must be finite and non-zero, every original diagonal entry
// positive.  Kept on the high words as integers, sign bit cleared (NaN and inf compare high and trip it).
struct BigGrowGuard {
    int hi = 0;
    __device__ __forceinline__ void see(const double prod) {
        const int h = __double2hiint(prod) & 0x7fffffff;
        hi = h > hi ? h : hi;
    }
    __device__ __forceinline__ bool bad(const double lim) const { return hi > __double2hiint(lim); }
};
struct BigPivGuard {
    int hi = 0;
    __device__ __forceinline__ void see(const double rinv) {          // 1 / 0 = inf, 1 / NaN = NaN
        const int h = __double2hiint(rinv) & 0x7fffffff;
        hi = h > hi ? h : hi;
    }
    __device__ __forceinline__ bool ok() const { return hi < 0x7ff00000; }
};

template <bool HL>
__device__ __forceinline__ double big_solve_diag(const DevModel& M, const BigWs& w, const int t, const int ka, const double g, bool& ok) {
    typedef double v4d __attribute__((ext_vector_type(4)));
    constexpr int NB = HL ? 16 : LU_NB;
    const int nr = M.nr;
    double* __restrict__ H = w.H;
    const int oU = HL ? w.ocu : NB * nr;        // dyn[oU + j * sU + c]: -U12(j, c-th trailing column)
    const int sU = HL ? nr : BT;
    auto HR = [&](const size_t i) -> double { if constexpr (HL) return dyn[i]; else return H[i]; };
    auto HW = [&](const size_t i, const double v) { if constexpr (HL) dyn[i] = v; else H[i] = v; };
    const int r = t;
    const bool row = r < nr;
    if (ka >= 0) lu_b[ka] = -g;
    const double d0 = row ? HR((size_t)r * nr + r) : 1.0;          // the scale of row r in the growth guard
    BigGrowGuard gg;
    BigPivGuard pg;
    __syncthreads();
    for (int kb = 0; kb < nr; kb += NB) {
        const int nb = nr - kb < NB ? nr - kb : NB;
        const int pb = HL ? kb * nr : 0;            // panel column j, row r: dyn[pb + j * nr + r]
        const int c0 = kb + nb;
        const int ncol = nr - c0;
        const bool live = row && r >= kb;
        const bool below = row && r >= c0;
        const bool own = (t >> 6) == (kb >> 6);     // this wavefront holds the diagonal block
        PROF_T0();
        double a[NB], u[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            a[j] = 0.0;
            if (live && j < nb) {
                if constexpr (HL) a[j] = dyn[pb + j * nr + r];
                else a[j] = H[(size_t)(kb + j) * nr + r];
            }
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) u[j] = (below && j < nb) ? HR((size_t)r * nr + kb + j) : 0.0;
        // NB == 16: the block's rows are one 16-lane DPP row, and the pivot row's broadcast rides on the FMA (v_fmac_f64_dpp row_newbcast,
        // rmx_device.h fmsub_rowbcast: half the issue slots of two v_readlane plus an FMA).  The wavefront's other rows cannot follow on
        // those instructions (a DPP row sees its own lanes): they take the LDS route of phase B with the other wavefronts.
        const bool own16 = NB == 16 && (t >> 4) == (kb >> 4);
        const bool inA = NB == 16 ? own16 : own;     // this thread's row is eliminated in phase A
        if constexpr (NB == 16) { if (own) {
            const int lb = kb & 63;
            const int i = t & 15;
            const bool mine = own16 && live;         // (a ragged last block: rows >= nb do not exist)
            double bb = mine ? lu_b[r] : 0.0, myrinv = 0.0;
#ifdef RMX_BIG_PROFILE
            const unsigned long long pf0 = __builtin_amdgcn_s_memtime();
#endif
            double rinv = recip(readlane_d(a[0], lb));
            static_for<16>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                constexpr bool NOPS = j >= 12;       // the last steps are short: a broadcast may follow the write of its source by < 2 VALU
                if (j < nb) {                        // workgroup-uniform
                    pg.see(rinv);
                    const bool el = mine && i > j;
                    const double l = el ? a[j] * rinv : 0.0;
                    gg.see(a[j] * l);
                    double rnext = 0.0;
                    if constexpr (j + 1 < 16) {
                        fmsub_rowbcast<j, NOPS>(a[j + 1], a[j + 1], l);
                        rnext = recip(readlane_d(a[j + 1], lb + j + 1));
                    }
#pragma unroll
                    for (int c = j + 2; c < 16; ++c) fmsub_rowbcast<j, NOPS>(a[c], a[c], l);
                    fmsub_rowbcast<j, NOPS>(bb, bb, l);
                    asm volatile("" : "+v"(rnext));  // formed HERE, among the updates (left alone it sinks into the next step's block)
                    if (el) a[j] = l;
                    if (i == j) myrinv = rinv;
                    rinv = rnext;
                }
            });
#ifdef RMX_BIG_PROFILE
            if (t == 0 && blockIdx.x == 0) g_prof[7] += __builtin_amdgcn_s_memtime() - pf0;
#endif
            if (mine) {
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    if (j < nb) dyn[pb + j * nr + r] = a[j];
                lu_b[r] = bb;
                lu_xs[r] = myrinv;
            }
        } }
        if (NB != 16 && own) {
            const int lb = kb & 63;
            const int i = (t & 63) - lb;             // row within the block (the wavefront's other rows: < 0 or >= nb)
            double bb = live ? lu_b[r] : 0.0, myrinv = 0.0;
#ifdef RMX_BIG_PROFILE
            const unsigned long long pf0 = __builtin_amdgcn_s_memtime();
#endif
            // The reciprocal of pivot j + 1 is started as soon as its column has been updated, underneath the updates of the other
            // columns (v_rcp_f64 and two Newton steps are ~100 ticks of dependent latency, the serial chain of a block otherwise)
            double rinv = recip(readlane_d(a[0], lb));
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                if (j < nb) {                        // workgroup-uniform
                    pg.see(rinv);
                    const bool el = live && i > j;
                    const double l = el ? a[j] * rinv : 0.0;
                    gg.see(a[j] * l);
                    double rnext = 0.0;
                    if (j + 1 < NB) {
                        const int j1 = j + 1 < NB ? j + 1 : j;
                        a[j1] = fma(-l, readlane_d(a[j1], lb + j), a[j1]);
                        rnext = recip(readlane_d(a[j1], lb + j1));
                    }
#pragma unroll
                    for (int c = j + 2; c < NB; ++c) a[c] = fma(-l, readlane_d(a[c], lb + j), a[c]);
                    bb = fma(-l, readlane_d(bb, lb + j), bb);
                    asm volatile("" : "+v"(rnext));  // formed HERE, among the updates (left alone it sinks into the next step's block)
                    if (el) a[j] = l;
                    if (i == j) myrinv = rinv;
                    rinv = rnext;
                }
            }
#ifdef RMX_BIG_PROFILE
            if (t == 0 && blockIdx.x == 0) g_prof[7] += __builtin_amdgcn_s_memtime() - pf0;
#endif
            if (live) {
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    if (j < nb) dyn[pb + j * nr + r] = a[j];
                if constexpr (!HL) {                 // U11 for the back substitution (the multipliers are never read from H again)
                    if (i < nb) {
#pragma unroll
                        for (int j = 0; j < NB; ++j)
                            if (j < nb && i <= j) H[(size_t)(kb + j) * nr + r] = a[j];
                    }
                }
                lu_b[r] = bb;
                if (i < nb) lu_xs[r] = myrinv;
            }
        }
        __syncthreads();
        PROF_ADD(3);
        if (ncol > 0) {          // a full panel (nb == NB) with rows and columns behind it
            PROF_T0();
            if constexpr (HL && !RMX_BIG_PHASEB_DPP_HL) {
            if (below) {
                if (!inA) {      // row r of L21: the rows of U11 come out of LDS as broadcasts
                    double bb = lu_b[r];
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        const double l = a[j] * lu_xs[kb + j];
                        gg.see(a[j] * l);
#pragma unroll
                        for (int c = j + 1; c < NB; ++c) a[c] = fma(-l, dyn[pb + c * nr + kb + j], a[c]);
                        bb = fma(-l, lu_b[kb + j], bb);
                        a[j] = l;
                    }
                    lu_b[r] = bb;
#pragma unroll
                    for (int j = 0; j < NB; ++j) dyn[pb + j * nr + r] = a[j];
                }
                // column c = r of U12: forward substitution with the unit lower triangle L11, a column of it per step
                const size_t col = (size_t)r * nr + kb;
#pragma unroll
                for (int i = 0; i < NB; ++i) {
#pragma unroll
                    for (int j = i + 1; j < NB; ++j) u[j] = fma(-dyn[pb + i * nr + kb + j], u[i], u[j]);
                    HW(col + i, u[i]);
                    dyn[oU + i * sU + (t - c0)] = -u[i];
                }
            }
            } else {
            // The entries of U11 / L11 are wavefront-uniform, and fetching each one as an LDS broadcast made this phase the LDS pipe's:
            // ~1 000 reads per thread and 32-column panel, ~8 clocks each with four wavefronts on one pipe.  Instead ROW j of U11 (column
            // i of L11) is read ONCE into a register - lane 16 r + c holds entry c, the same in all four 16-lane rows - and every
            // update takes its entry from there on the DPP broadcast that rides on the FMA (fmsub_rowbcast): 16 / 64 reads instead of
            // 240 / 992.  DPP reads lanes whatever their row does, so the whole wavefront runs the loop (rows that take no part carry the
            // multiplier 0), under a wavefront-uniform condition.
            if ((t | 63) >= c0 && (t & ~63) < nr) {
                const int l15 = t & 15;
                const bool partL = below && !inA;
                if (!(NB == 32 && own)) {        // (32-column panels: the owner wavefront eliminated all of its rows in phase A)
                    double bb = partL ? lu_b[r] : 0.0;
                    static_for<NB>([&](auto jc) {
                        constexpr int j = decltype(jc)::value;
                        const double ulo = dyn[pb + l15 * nr + kb + j];
                        double uhi = 0.0;
                        if constexpr (NB == 32) uhi = dyn[pb + (16 + l15) * nr + kb + j];
                        const double l = partL ? a[j] * lu_xs[kb + j] : 0.0;
                        gg.see(a[j] * l);
                        static_for<NB>([&](auto cc) {
                            constexpr int c = decltype(cc)::value;
                            if constexpr (c > j && c < 16) fmsub_rowbcast<c>(a[c], ulo, l);
                            if constexpr (c > j && c >= 16) fmsub_rowbcast<c - 16>(a[c], uhi, l);
                        });
                        bb = fma(-l, lu_b[kb + j], bb);
                        if (partL) a[j] = l;
                    });
                    if (partL) {
                        lu_b[r] = bb;
#pragma unroll
                        for (int j = 0; j < NB; ++j) dyn[pb + j * nr + r] = a[j];
                    }
                }
                // column c = r of U12: forward substitution with the unit lower triangle L11, a column of it per step
                const size_t col = (size_t)r * nr + kb;
                static_for<NB>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    const double llo = dyn[pb + i * nr + kb + l15];
                    double lhi = 0.0;
                    if constexpr (NB == 32) lhi = dyn[pb + i * nr + kb + 16 + l15];
                    static_for<NB>([&](auto jc) {
                        constexpr int j = decltype(jc)::value;
                        if constexpr (j > i && j < 16) fmsub_rowbcast<j>(u[j], llo, u[i]);
                        if constexpr (j > i && j >= 16) fmsub_rowbcast<j - 16>(u[j], lhi, u[i]);
                    });
                    if (below) {
                        HW(col + i, u[i]);
                        dyn[oU + i * sU + (t - c0)] = -u[i];
                    }
                });
            }
            }
            __syncthreads();
            PROF_ADD(4);
            // A22 -= L21 U12 on the matrix cores: the tiling of big_solve_blocked, over the row blocks from c0 on
            {
                PROF_T0();
                const int wave = t >> 6, lane = t & 63, jj = lane & 15, gg4 = lane >> 4;
                const int MB = (nr + 15) >> 4, NBK = (ncol + 15) >> 4;
                for (int mb = (c0 >> 4) + wave; mb < MB; mb += BT / 64) {
                    const int arow = 16 * mb + jj;
                    const bool aon = arow < nr;
                    double lf[NB / 4];
#pragma unroll
                    for (int kk = 0; kk < NB / 4; ++kk) lf[kk] = aon ? dyn[pb + (4 * kk + gg4) * nr + arow] : 0.0;
                    // C two tiles ahead: a tile in flight per wavefront is 2 KB, four wavefronts 8 KB per CU - at the ~1 us of a trip to
                    // HBM / MALL that is far from what the CU's share of the bandwidth can carry
                    struct Tile { v4d c; size_t ad[4]; bool ok[4]; };
                    auto fetch = [&](const int nbk) {
                        Tile T;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int cc = 16 * nbk + 4 * q + gg4;
                            T.ok[q] = cc < ncol && aon;
                            T.ad[q] = (size_t)(c0 + (cc < ncol ? cc : 0)) * nr + (aon ? arow : 0);
                            T.c[q] = T.ok[q] ? HR(T.ad[q]) : 0.0;
                        }
                        return T;
                    };
                    auto work = [&](const int nbk, const Tile& T) {
                        const int cj = 16 * nbk + jj;                    // this lane's column of the A operand
                        double uf[NB / 4];
#pragma unroll
                        for (int kk = 0; kk < NB / 4; ++kk) uf[kk] = cj < ncol ? dyn[oU + (4 * kk + gg4) * sU + cj] : 0.0;
                        v4d acc = T.c;
#pragma unroll
                        for (int kk = 0; kk < NB / 4; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(uf[kk], lf[kk], acc, 0, 0, 0);
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (T.ok[q]) HW(T.ad[q], acc[q]);
                    };
                    Tile TA = fetch(0), TB = fetch(1);                   // (a tile past the end: all four of its elements masked)
                    for (int nbk = 0; nbk < NBK; nbk += 2) {
                        const Tile T0 = TA, T1 = TB;
                        TA = fetch(nbk + 2);
                        TB = fetch(nbk + 3);
                        work(nbk, T0);
                        if (nbk + 1 < NBK) work(nbk + 1, T1);
                    }
                }
                __syncthreads();
                PROF_ADD(5);
            }
        }
    }
    // the guard, once per solve: l'^2 = |l a| / d_r <= LU_GROWTH_MAX^2 for every multiplier of row r, every pivot finite and non-zero
    const bool bad = (row && (!(d0 > 0.0) || gg.bad(LU_GROWTH_MAX * LU_GROWTH_MAX * d0))) || !pg.ok();
    ok = block_any(bad, t) == 0;
#ifdef RMX_BIG_PROFILE
    {
        const double ng = block_sum((row && gg.bad(LU_GROWTH_MAX * LU_GROWTH_MAX * d0)) ? 1.0 : 0.0, t);
        const double nd = block_sum((row && !(d0 > 0.0)) ? 1.0 : 0.0, t);
        const double np = block_sum(pg.ok() ? 0.0 : 1.0, t);
        if (!ok && t == 0 && blockIdx.x < 48) printf("guard tripped, block %d: rows over the growth limit %g, diagonal <= 0: %g, threads that saw a bad pivot %g\n", blockIdx.x, ng, nd, np);
    }
#endif
    // back substitution, ONE barrier per block of NB pivots: the wavefront that owns the block's rows solves the triangle in registers
    // (x_j broadcast with v_readlane, the other rows it owns updated on the same instructions) and publishes x; the rows of the other
    // wavefronts take the block's NB terms after the barrier, their entries of U already in flight.  (One barrier per pivot: 113 k ticks
    // of the 900 k of a solve at 256 DOFs.)
    PROF_T0();
    double bb = row ? lu_b[r] : 0.0;
    for (int kb = ((nr - 1) / NB) * NB; kb >= 0; kb -= NB) {
        const int nb = nr - kb < NB ? nr - kb : NB;
        const bool own = (t >> 6) == (kb >> 6);
        double uu[NB], ri[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) uu[j] = (row && j < nb && r < kb + j) ? HR((size_t)(kb + j) * nr + r) : 0.0;
        if (own) {
            const int lb = kb & 63;
#pragma unroll
            for (int j = 0; j < NB; ++j) ri[j] = j < nb ? lu_xs[kb + j] : 0.0;
#pragma unroll
            for (int j = NB - 1; j >= 0; --j) {
                if (j < nb) {                        // workgroup-uniform
                    const double xj = readlane_d(bb, lb + j) * ri[j];
                    bb = (r == kb + j) ? xj : fma(-uu[j], xj, bb);
                }
            }
            if (row && r >= kb && r < kb + nb) lu_b[r] = bb;
        }
        __syncthreads();
        if (!own && row && r < kb) {
#pragma unroll
            for (int j = 0; j < NB; ++j)
                if (j < nb) bb = fma(-uu[j], lu_b[kb + j], bb);
        }
    }
    const double dxr = ka >= 0 ? lu_b[ka] : 0.0;
    PROF_ADD(6);
    return dxr;
}

#if RMX_BIG_PIVOT_INLINE
#define BIG_PIVOT_INL __forceinline__
#else
#define BIG_PIVOT_INL __noinline__
#endif
template <bool HL>
__device__ BIG_PIVOT_INL double big_solve(const DevModel& M, const BigWs& w, const int t, const int ka, const double g) {
    if constexpr (!HL) return big_solve_blocked<false>(M, w, t, ka, g);
    // H in LDS: from ~100 DOFs up the blocked form (16-column panels in place, trailing update on the matrix cores) is ahead - the
    // unblocked update below moves the whole trailing matrix through LDS once per pivot, 1.5 k clocks of LDS bandwidth at 128 DOFs;
    // below that the per-column cost of a panel (~3 k ticks of serial search / reciprocal / look-ahead) outweighs it (72 DOFs:
    // 1.57 vs 1.77 ms per step; 128: 4.63 vs 4.40; profiles/r04u_*)
    if constexpr (HL) {
        if (M.nr >= 100) return big_solve_blocked<true>(M, w, t, ka, g);       // workgroup-uniform
    }
    const int nr = M.nr;
    double* __restrict__ H = w.H;
    if (ka >= 0) lu_b[ka] = -g;
    __syncthreads();
    // thread = (reduced row r, column group cg): the BT / nr threads of a row share its trailing columns (c = k+1+cg, step ncg); with
    // one thread per row the update of a row was nr - k dependent global round trips per pivot, which is where the time went
    const int ncg = BT / nr > 0 ? BT / nr : 1;
    const int r = t % nr, cg = t / nr;
    const bool row = cg < ncg;
    int mystep = -1;
    for (int k = 0; k < nr; ++k) {
        // pivot search over the unused rows: max |H(r,k)|, lowest row among equals
        const size_t ck = (size_t)k * nr;
        const double cand = (row && cg == 0 && mystep < 0) ? fabs(hget<HL>(H, ck + r)) : -1.0;
        const int pr = block_argmax(cand, r, t);
        if (t == 0) lu_piv[k] = pr;
        if (r == pr) mystep = k;
        if (row && mystep < 0) {
            const double l = hget<HL>(H, ck + r) * recip(hget<HL>(H, ck + pr));      // dgetf2 scales by the reciprocal of the pivot
            int c = k + 1 + cg;
            // UF columns in flight (loads first, then the stores): 4 when H is in LDS, 16 when every access is a trip to HBM / L2
            constexpr int UF = HL ? 4 : 16;
            for (; c + (UF - 1) * ncg < nr; c += UF * ncg) {
                double pv[UF], av[UF];
#pragma unroll
                for (int u = 0; u < UF; ++u) pv[u] = hget<HL>(H, (size_t)(c + u * ncg) * nr + pr);
#pragma unroll
                for (int u = 0; u < UF; ++u) av[u] = hget<HL>(H, (size_t)(c + u * ncg) * nr + r);
#pragma unroll
                for (int u = 0; u < UF; ++u) hput<HL>(H, (size_t)(c + u * ncg) * nr + r, av[u] - l * pv[u]);
            }
            if constexpr (!HL) {          // remainder of the wide unroll, four at a time
                for (; c + 3 * ncg < nr; c += 4 * ncg) {
                    double pv[4], av[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) pv[u] = hget<HL>(H, (size_t)(c + u * ncg) * nr + pr);
#pragma unroll
                    for (int u = 0; u < 4; ++u) av[u] = hget<HL>(H, (size_t)(c + u * ncg) * nr + r);
#pragma unroll
                    for (int u = 0; u < 4; ++u) hput<HL>(H, (size_t)(c + u * ncg) * nr + r, av[u] - l * pv[u]);
                }
            }
            for (; c < nr; c += ncg) {
                const size_t cc = (size_t)c * nr;
                hput<HL>(H, cc + r, hget<HL>(H, cc + r) - l * hget<HL>(H, cc + pr));
            }
            if (cg == 0) lu_b[r] -= l * lu_b[pr];
        }
        __syncthreads();
    }
    // back substitution on the implicitly permuted upper triangle (one barrier per step: see big_solve_blocked)
    if (t < nr) lu_xs[t] = recip(hget<HL>(H, (size_t)t * nr + lu_piv[t]));
    __syncthreads();
    double dxr = 0.0;
    for (int k = nr - 1; k >= 0; --k) {
        const double xk = lu_b[lu_piv[k]] * lu_xs[k];
        if (k == ka) dxr = xk;
        if (row && cg == 0 && mystep < k) lu_b[r] -= hget<HL>(H, (size_t)k * nr + r) * xk;
        __syncthreads();
    }
    return dxr;
}

// newton (driverRedMaxBDF1.m:94-157) for one implicit solve; see newton_impl (rmx_device.h) for the stall shortcut and the
// compensated iterate x + lo.  Every decision is workgroup-uniform (norms come out of block_sum identical in all threads).
template <bool HL, bool CT>
__device__ __forceinline__ double big_newton_inl(const DevModel& M, const DevOpts& o, const BigWs& w, const NodeConsts& nc, const int t,
                                                 const int ka, double x, const double qA, const double qB, const double eta, BigOut& last,
                                                 int& iters, int& halvings, int& status, double& xlo) {
#if RMX_BIG_NEWTON_ROT
    // The ROTATED loop (newton_rot of rmx_device.h): the reference evaluates the residual twice at every point it accepts - as the line
    // search's trial, then with H at the top of the next iteration.  Here the evaluation at the top of the loop IS the first trial of the
    // running line search, taken with H: accepted - the usual case - the solve follows at once and a residual-only evaluation (57 of
    // 361 k ticks per iteration at 128 DOFs) is saved.  The evaluation asks back once its residual is complete (the gate of big_eval):
    // a trial that is rejected, or accepted and converged, ends there, before the Hessian's share of the work - nothing is wasted; the
    // halving then goes on with residual-only evaluations and H is evaluated at the point it ends on, as before.  Same points, same tests,
    // same decisions as the loop below (g of the two instantiations agrees bit for bit, tests/test_gpu_big_trees.py); ONE call site for
    // the evaluation with H, one for the residual-only one (the layout finding above).
    double lo = 0.0;
    BigOut e, e0;
    int iter = 1, lsfail = 0, pivstreak = 0, pivhold = 0, iterLs = 1;
    bool inLs = false, repivot = false, stalled = false;
    double dx = 0.0, alpha = 1.0, g0n2 = 0.0, f0 = 0.0, gn2 = 0.0, x0 = x, lo0 = 0.0;
    while (true) {
        bool hvalid = true;
        PROF_T0();
        big_eval<true, HL, CT>(M, w, nc, t, x, ((x - qA) + lo) / eta, (x - qB) + lo, eta, e, [&](const double g) {
            if (!inLs) return false;
            gn2 = block_sum(g * g, t);
            const bool accepted = 0.5 * gn2 < f0;
            // no H for: a rejected trial with halvings left; a point the Newton loop ends on (converged, or out of iterations)
            hvalid = accepted ? !(sqrt(gn2) < o.tol || iter >= o.iterMax) : !(iterLs < o.iterLsMax || sqrt(gn2) < o.tol || iter >= o.iterMax);
            return !hvalid;
        });
        PROF_ADD(0);
        last = e;
        if (inLs) {               // (x, lo) is the first trial point of the line search from (x0, lo0) along dx; gn2 = |g|^2 there
            inLs = false;
            if (!(0.5 * gn2 < f0)) {
                while (true) {
                    if (iterLs >= o.iterLsMax) break;
                    alpha *= 0.5;
                    ++iterLs;
                    two_sum(x0, fma(alpha, dx, lo0), x, lo);
                    lo *= o.comp;
                    if (block_all(x == x0 && lo == lo0, t)) {
                        stalled = true;
                        iterLs = o.iterLsMax;
                        e = e0;
                        break;
                    }
                    { PROF_T0(); big_eval<false, false, CT>(M, w, nc, t, x, ((x - qA) + lo) / eta, (x - qB) + lo, eta, e); PROF_ADD(2); }
                    hvalid = false;
                    gn2 = block_sum(e.g * e.g, t);
                    if (0.5 * gn2 < f0) break;
                }
            }
            last = e;
            halvings += iterLs - 1;
            if (stalled) {
                if (!(sqrt(g0n2) < o.tol)) status |= 2 | 8;
                break;
            }
            if (sqrt(gn2) < o.tol) break;
            if (iter >= o.iterMax) { status |= 2; break; }
            lsfail += (0.5 * gn2 < f0) ? 0 : 1;              // rmx_opts.ls_fail_limit, see newton_impl
            if (o.lsFailLimit > 0 && lsfail >= o.lsFailLimit) { status |= 2 | ST_LS_CUT; break; }
            ++iter;
            if (!hvalid) continue;                        // H at the accepted point: the evaluation at the top
        }
        // e: g and H at (x, lo).  Elimination on the diagonal; if its guard trips, H again (the solve destroyed it in place) and pivoting
        if (!repivot) ++iters;
        if (!repivot && o.lu_mode == 0 && pivhold == 0) {     // workgroup-uniform
            bool lu_ok = false;
            { PROF_T0(); dx = big_solve_diag<HL>(M, w, t, ka, e.g, lu_ok); PROF_ADD(1); }
            if (!lu_ok) {
                status |= 16;                                // growth guard tripped
                if (++pivstreak >= 2) pivhold = 1;           // a solve that keeps tripping: partial pivoting for the rest of this solve
                repivot = true;
                continue;
            }
            pivstreak = 0;
        } else {
            { PROF_T0(); dx = big_solve<HL>(M, w, t, ka, e.g); PROF_ADD(1); }
            repivot = false;
        }
        e0 = e;
        const double dxn2 = block_sum(dx * dx, t);
        if (!(dxn2 == dxn2)) { status |= 4; break; }
        if (sqrt(dxn2) > o.dxMax) { status |= 1; break; }
        alpha = 1.0;
        g0n2 = block_sum(e.g * e.g, t);
        f0 = 0.5 * g0n2;
        x0 = x;
        lo0 = lo;
        iterLs = 1;
        gn2 = g0n2;
        two_sum(x0, fma(alpha, dx, lo0), x, lo);
        lo *= o.comp;
        if (block_all(x == x0 && lo == lo0, t)) {          // the first trial is the point itself: the stall shortcut of newton_impl
            last = e;
            halvings += o.iterLsMax - 1;
            if (!(sqrt(g0n2) < o.tol)) status |= 2 | 8;
            break;
        }
        inLs = true;
    }
#else
    double lo = 0.0;
    BigOut e;
    int iter = 1, lsfail = 0, pivstreak = 0, pivhold = 0;
    while (true) {
#if RMX_BIG_NEWTON_ONE_SITE
        // One call site each for the evaluation and the two solves (all three are inlined: a call costs these kernels more than it
        // saves - arguments by reference live in scratch, the callee saves its registers).  Round 0: H, then elimination on the diagonal;
        // if its guard trips, round 1: H again (the solve destroyed it in place), then partial pivoting.
        double dx = 0.0;
#pragma unroll 1
        for (int round = 0; round < 2; ++round) {
            { PROF_T0(); big_eval<true, HL, CT>(M, w, nc, t, x, ((x - qA) + lo) / eta, (x - qB) + lo, eta, e); PROF_ADD(0); }
            PROF_T0();
            if (round == 0 && o.lu_mode == 0 && pivhold == 0) {          // workgroup-uniform
                bool lu_ok = false;
                dx = big_solve_diag<HL>(M, w, t, ka, e.g, lu_ok);
                PROF_ADD(1);
                if (lu_ok) {
                    pivstreak = 0;
                    break;
                }
                status |= 16;                                // growth guard tripped
                if (++pivstreak >= 2) pivhold = 1;           // a solve that keeps tripping: partial pivoting for the rest of this solve
            } else {
                dx = big_solve<HL>(M, w, t, ka, e.g);
                PROF_ADD(1);
                break;
            }
        }
        const BigOut e0 = e;
        last = e;
        ++iters;
#else
        { PROF_T0(); big_eval<true, HL, CT>(M, w, nc, t, x, ((x - qA) + lo) / eta, (x - qB) + lo, eta, e); PROF_ADD(0); }
        const BigOut e0 = e;
        last = e;
        ++iters;
        double dx;
        {
            PROF_T0();
            bool lu_ok = false;
            if (o.lu_mode == 0 && pivhold == 0) dx = big_solve_diag<HL>(M, w, t, ka, e.g, lu_ok);
            if (!lu_ok) {
                if (o.lu_mode == 0 && pivhold == 0) {        // growth guard tripped: H was destroyed in place - re-assemble, then pivot
                    status |= 16;
                    if (++pivstreak >= 2) pivhold = 1;       // a solve that keeps tripping: partial pivoting for the rest of this solve
                    big_eval<true, HL, CT>(M, w, nc, t, x, ((x - qA) + lo) / eta, (x - qB) + lo, eta, e);
                }
                dx = big_solve<HL>(M, w, t, ka, e.g);
            } else {
                pivstreak = 0;
            }
            PROF_ADD(1);
        }
#endif
        const double dxn2 = block_sum(dx * dx, t);
        if (!(dxn2 == dxn2)) { status |= 4; break; }
        if (sqrt(dxn2) > o.dxMax) { status |= 1; break; }
        double alpha = 1.0;
        const double g0n2 = block_sum(e.g * e.g, t);
        const double f0 = 0.5 * g0n2;
        const double x0 = x, lo0 = lo;
        int iterLs = 1;
        double gn2 = g0n2;
        bool stalled = false;
        while (true) {
            two_sum(x0, fma(alpha, dx, lo0), x, lo);
            lo *= o.comp;
            if (block_all(x == x0 && lo == lo0, t)) {
                stalled = true;
                iterLs = o.iterLsMax;
                e = e0;
                break;
            }
            { PROF_T0(); big_eval<false, false, CT>(M, w, nc, t, x, ((x - qA) + lo) / eta, (x - qB) + lo, eta, e); PROF_ADD(2); }
            gn2 = block_sum(e.g * e.g, t);
            if (0.5 * gn2 < f0) break;
            if (iterLs >= o.iterLsMax) break;
            alpha *= 0.5;
            ++iterLs;
        }
        last = e;
        halvings += iterLs - 1;
        if (stalled) {
            if (!(sqrt(g0n2) < o.tol)) status |= 2 | 8;
            break;
        }
        if (sqrt(gn2) < o.tol) break;
        if (iter >= o.iterMax) { status |= 2; break; }
        lsfail += (0.5 * gn2 < f0) ? 0 : 1;              // rmx_opts.ls_fail_limit, see newton_impl
        if (o.lsFailLimit > 0 && lsfail >= o.lsFailLimit) { status |= 2 | ST_LS_CUT; break; }
        ++iter;
    }
#endif
    xlo = lo;
    return x;
}
// The BDF2 kernels solve at three places (two SDIRK2 stages, the BDF2 step): one out-of-line copy; the BDF1 kernels inline theirs.
template <bool HL, bool CT>
__device__ __noinline__ double big_newton(const DevModel& M, const DevOpts& o, const BigWs& w, const NodeConsts& nc, const int t, const int ka,
                                          double x, const double qA, const double qB, const double eta, BigOut& last, int& iters, int& halvings,
                                          int& status, double& xlo) {
    return big_newton_inl<HL, CT>(M, o, w, nc, t, ka, x, qA, qB, eta, last, iters, halvings, status, xlo);
}

// Joint.reparam -> JointSpherical.reparam_ for every spherical group (sph_reparam in rmx_device.h, with the group's values read from
// the workspace instead of v_readlane).  Returns true if a chart changed (the caller refreshes its NodeConsts).
template <bool WITH_PREV>
__device__ bool big_reparam(const DevModel& M, const BigWs& w, const int t, int* chart, double& q, double& qd, double& qp, double& qdp) {
    const int LS = w.ns;         // q, qd, qp, qdp per node at the start of region X: [4][ns]
    __syncthreads();
    if (t < LS) {
        dyn[t] = q;
        dyn[LS + t] = qd;
        dyn[2 * LS + t] = qp;
        dyn[3 * LS + t] = qdp;
    }
    __syncthreads();
    bool switched = false;
    for (int g = 0; g < M.nsph; ++g) {
        const int first = M.sph_first[g];
        const int c0 = chart[g];
        const double qv[3] = {dyn[first], dyn[first + 1], dyn[first + 2]};
        double Told[9];
        const double detTold = euler_T(c0, qv, Told);
        if (fabs(detTold) > 0.5) continue;
        const double qdv[3] = {dyn[LS + first], dyn[LS + first + 1], dyn[LS + first + 2]};
        const double q1v[3] = {dyn[2 * LS + first], dyn[2 * LS + first + 1], dyn[2 * LS + first + 2]};
        const double qd1v[3] = {dyn[3 * LS + first], dyn[3 * LS + first + 1], dyn[3 * LS + first + 2]};
        double R[9], R1[9], Tt[9];
        euler_R(c0, qv, R);
        if (WITH_PREV) euler_R(c0, q1v, R1);
        int best = 1;
        double bestv = -1.0;
        for (int k = 1; k <= 12; ++k) {
            double qk[3];
            euler_inv(k, R, qk);
            double vv = fabs(euler_T(k, qk, Tt));
            vv = (vv == vv) ? vv : 0.0;
            if (WITH_PREV) {
                euler_inv(k, R1, qk);
                double v1 = fabs(euler_T(k, qk, Tt));
                v1 = (v1 == v1) ? v1 : 0.0;
                vv = v1 < vv ? v1 : vv;
            }
            if (vv > bestv) {
                bestv = vv;
                best = k;
            }
        }
        double wv[3], Tn[9], qn[3], qdn[3], q1n[3] = {0, 0, 0}, qd1n[3] = {0, 0, 0};
#pragma unroll
        for (int i = 0; i < 3; ++i) wv[i] = Told[3 * i] * qdv[0] + Told[3 * i + 1] * qdv[1] + Told[3 * i + 2] * qdv[2];
        euler_inv(best, R, qn);
        euler_T(best, qn, Tn);
        solve3(Tn, wv, qdn);
        if (WITH_PREV) {
            euler_T(c0, q1v, Told);
#pragma unroll
            for (int i = 0; i < 3; ++i) wv[i] = Told[3 * i] * qd1v[0] + Told[3 * i + 1] * qd1v[1] + Told[3 * i + 2] * qd1v[2];
            euler_inv(best, R1, q1n);
            euler_T(best, q1n, Tn);
            solve3(Tn, wv, qd1n);
        }
        for (int k = 0; k < 3; ++k)
            if (t == first + k) {
                q = qn[k];
                qd = qdn[k];
                if (WITH_PREV) {
                    qp = q1n[k];
                    qdp = qd1n[k];
                }
            }
        if (best != c0) switched = true;
        __syncthreads();                 // every thread has read chart[g]
        if (best != c0 && t == 0) chart[g] = best;
    }
    __syncthreads();
    return switched;
}

// simLoop of driverRedMaxBDF1.m:57-91 (INTEG 1) / driverRedMaxBDF2.m:57-125 (INTEG 2), all steps inside one launch
template <int INTEG, bool HL, bool CT>
__global__ void __launch_bounds__(BT) k_big_step(const DevModel M, const DevOpts o, const StepArgs a, double* wsbase, const size_t wsstride) {
    const int t = threadIdx.x, traj = blockIdx.x;
    const unsigned long long tick0 = __builtin_amdgcn_s_memtime();
    const BigWs w = big_ws<HL>(wsbase + (size_t)traj * wsstride, M.n, M.nr, M.con != nullptr);
    const int id = (t < M.n) ? M.idx[t] : -1;
    const size_t off = (size_t)traj * M.nr + (id >= 0 ? id : 0);
    double q = id >= 0 ? a.q[off] : 0.0;
    double qd = id >= 0 ? a.qd[off] : 0.0;
    double qp = (INTEG == 2 && id >= 0) ? a.qp[off] : 0.0;
    double qdp = (INTEG == 2 && id >= 0) ? a.qdp[off] : 0.0;
    const bool started = INTEG == 2 && (*a.started) != 0;
    int* const chart = M.nsph ? a.chart + (size_t)traj * M.nsph : nullptr;
    NodeConsts nc = node_consts(M, t < M.n ? t : 0, chart);
    const double h = o.h;
    int iters = 0, halv = 0, status = 0;
    for (int s = 0; s < a.nsteps; ++s) {
        BigOut last;
        double xlo;
        if (INTEG == 1) {
            const double q0 = q, qd0 = qd;
            const double xg = q0 + h * qd0;
            const double x = big_newton_inl<HL, CT>(M, o, w, nc, t, id, xg, q0, xg, h, last, iters, halv, status, xlo);
            qd = ((x - q0) + xlo) / h;
            q = x;
        } else if (s == 0 && !started) {
            const double al = (2.0 - sqrt(2.0)) / 2.0;
            const double q0 = q, qd0 = qd;
            const double qa = big_newton<HL, CT>(M, o, w, nc, t, id, q0 + al * h * qd0, q0, q0 + (al * h) * qd0, al * h, last, iters, halv, status, xlo);
            const double qda = (qa - q0) / (al * h);
            const double x10 = qa + (1.0 - al) * h * qda;
            const double qA = q0 + (1.0 - al) * h * qda;
            const double qB = q0 + (2.0 * al - 1.0) * h * qd0 + 2.0 * (1.0 - al) * h * qda;
            const double q1 = big_newton<HL, CT>(M, o, w, nc, t, id, x10, qA, qB, al * h, last, iters, halv, status, xlo);
            qd = (q1 - q0 - (1.0 - al) * h * qda) / (al * h);
            q = q1;
            qp = q0;
            qdp = qd0;
        } else {
            const double q0 = qp, qd0 = qdp, q1 = q, qd1 = qd;
            const double x0 = q1 + h * qd1;
            const double qA = (4.0 / 3.0) * q1 - (1.0 / 3.0) * q0;
            const double qB = (4.0 / 3.0) * q1 - (1.0 / 3.0) * q0 + (8.0 / 9.0) * h * qd1 - (2.0 / 9.0) * h * qd0;
            const double q2 = big_newton<HL, CT>(M, o, w, nc, t, id, x0, qA, qB, (2.0 / 3.0) * h, last, iters, halv, status, xlo);
            qp = q1;
            qdp = qd1;
            qd = (3.0 / (2.0 * h)) * (q2 - (4.0 / 3.0) * q1 + (1.0 / 3.0) * q0);
            q = q2;
        }
        if (M.nsph) {      // jroot.reparam() (driverRedMaxBDF1.m:78, driverRedMaxBDF2.m:112)
            double np0 = 0.0, np1 = 0.0;
            const bool sw = INTEG == 1 ? big_reparam<false>(M, w, t, chart, q, qd, np0, np1) : big_reparam<true>(M, w, t, chart, q, qd, qp, qdp);
            if (sw) {
                status |= 32;
                nc = node_consts(M, t < M.n ? t : 0, chart);
            }
        }
        if (a.histT) {
            const double T = block_sum(last.eT, t), V = block_sum(last.eV, t);
            if (t == 0) {
                a.histT[(size_t)s * a.B + traj] = T;
                a.histV[(size_t)s * a.B + traj] = V;
            }
        }
        if (a.histQ && id >= 0) {
            a.histQ[(size_t)s * a.B * M.nr + off] = q;
            a.histQd[(size_t)s * a.B * M.nr + off] = qd;
        }
        if (a.histC && t < M.nsph) a.histC[((size_t)s * a.B + traj) * M.nsph + t] = chart[t];
    }
    if (id >= 0) {
        a.q[off] = q;
        a.qd[off] = qd;
        if (INTEG == 2) {
            a.qp[off] = qp;
            a.qdp[off] = qdp;
        }
    }
    if (t == 0 && a.it) {
        a.it[traj] += iters;
        a.ls[traj] += halv;
        a.status[traj] |= status;
    }
    if (t == 0 && a.ticks) a.ticks[traj] += __builtin_amdgcn_s_memtime() - tick0;
#ifdef RMX_BIG_PROFILE
    if (t == 0 && traj == 0)
        printf("big profile (ticks, block 0): eval+H %llu solve %llu trial eval %llu | panel %llu U12 %llu trailing %llu backsub %llu | diagonal blocks of wavefront 0 %llu | total %llu\n",
               g_prof[0], g_prof[1], g_prof[2], g_prof[3], g_prof[4], g_prof[5], g_prof[6], g_prof[7], __builtin_amdgcn_s_memtime() - tick0);
#endif
}

// Parity hook (rmx_eval): one residual (+ Hessian) evaluation per trajectory
template <bool WANT_H, bool HL, bool CT>
__global__ void __launch_bounds__(BT) k_big_eval(const DevModel M, const double* __restrict__ q, const double* __restrict__ qA, const double* __restrict__ qB,
                                                 const double eta, double* __restrict__ g, double* __restrict__ H, const int* __restrict__ charts,
                                                 double* wsbase, const size_t wsstride) {
    const int t = threadIdx.x, traj = blockIdx.x;
    const BigWs w = big_ws<HL>(wsbase + (size_t)traj * wsstride, M.n, M.nr, M.con != nullptr);
    const int id = (t < M.n) ? M.idx[t] : -1;
    const size_t off = (size_t)traj * M.nr + (id >= 0 ? id : 0);
    const double x = id >= 0 ? q[off] : 0.0, xa = id >= 0 ? qA[off] : 0.0, xb = id >= 0 ? qB[off] : 0.0;
    const NodeConsts nc = node_consts(M, t < M.n ? t : 0, M.nsph ? charts + (size_t)traj * M.nsph : nullptr);
    BigOut e;
    big_eval<WANT_H, HL, CT>(M, w, nc, t, x, (x - xa) / eta, x - xb, eta, e);
    if (id >= 0) g[off] = e.g;
    if (WANT_H) {
        const size_t nn = (size_t)M.nr * M.nr;
        for (size_t i = t; i < nn; i += BT) H[(size_t)traj * nn + i] = hget<HL>(w.H, i);
    }
}

// Joint.computeEnergies / Body.computeEnergies at the stored state
__global__ void __launch_bounds__(BT) k_big_energy(const DevModel M, const double* __restrict__ q, const double* __restrict__ qd, double* __restrict__ T,
                                                   double* __restrict__ V, const int* __restrict__ charts, double* wsbase, const size_t wsstride) {
    const int t = threadIdx.x, traj = blockIdx.x;
    const BigWs w = big_ws<false>(wsbase + (size_t)traj * wsstride, M.n, M.nr, M.con != nullptr);
    const int id = (t < M.n) ? M.idx[t] : -1;
    const size_t off = (size_t)traj * M.nr + (id >= 0 ? id : 0);
    const NodeConsts nc = node_consts(M, t < M.n ? t : 0, M.nsph ? charts + (size_t)traj * M.nsph : nullptr);
    BigOut e;
    if (M.con) big_eval<false, false, true>(M, w, nc, t, id >= 0 ? q[off] : 0.0, id >= 0 ? qd[off] : 0.0, 0.0, 1.0, e);
    else big_eval<false>(M, w, nc, t, id >= 0 ? q[off] : 0.0, id >= 0 ? qd[off] : 0.0, 0.0, 1.0, e);
    const double tt = block_sum(e.eT, t), vv = block_sum(e.eV, t);
    if (t == 0) {
        T[traj] = tt;
        V[traj] = vv;
    }
}

}  // namespace

size_t big_ws_doubles(const rmx_model* m) { return big_ws_doubles_n(m->nr); }

// Dynamic LDS of a launch: the per-node workspace, plus H when nr x nr doubles fit the workgroup's limit next to it and ~8 KB of
// static arrays (hl)
static bool big_hl(const rmx_model* m) {
    return big_lds_doubles(m->n, m->nr, true, m->dm.con != nullptr) * sizeof(double) + 8192 <= (size_t)m->lds_limit;      // the kernels hold 6.1 - 7.1 KB of static LDS
}
static size_t big_dyn_lds(const rmx_model* m, const bool hl) { return big_lds_doubles(m->n, m->nr, hl, m->dm.con != nullptr) * sizeof(double); }
template <typename K>
static void big_allow_lds(K kernel, const size_t bytes) {
    if (bytes > 48 * 1024) (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}
template <int INTEG, bool HL, bool CT>
static void big_step_launch(const rmx_model* m, const rmx_batch* b, const DevOpts& o, const StepArgs& a, const size_t lds) {
    big_allow_lds(k_big_step<INTEG, HL, CT>, lds);
    k_big_step<INTEG, HL, CT><<<dim3(b->B), dim3(BT), lds, b->stream>>>(m->dm, o, a, b->bigws, b->bigws_stride);
}
void launch_big_step(const rmx_model* m, const rmx_batch* b, int integ, const DevOpts& o, const StepArgs& a) {
    b->last_kernel = "k_big_step";
    const bool hl = big_hl(m), ct = m->dm.con != nullptr, b1 = integ == INTEG_BDF1;
    const size_t lds = big_dyn_lds(m, hl);
    if (ct) {
        if (hl) { if (b1) big_step_launch<1, true, true>(m, b, o, a, lds); else big_step_launch<2, true, true>(m, b, o, a, lds); }
        else { if (b1) big_step_launch<1, false, true>(m, b, o, a, lds); else big_step_launch<2, false, true>(m, b, o, a, lds); }
    } else {
        if (hl) { if (b1) big_step_launch<1, true, false>(m, b, o, a, lds); else big_step_launch<2, true, false>(m, b, o, a, lds); }
        else { if (b1) big_step_launch<1, false, false>(m, b, o, a, lds); else big_step_launch<2, false, false>(m, b, o, a, lds); }
    }
}
template <bool WANT_H, bool HL, bool CT>
static void big_eval_launch(const rmx_model* m, const rmx_batch* b, double eta, double* dg, double* dH, const size_t lds) {
    big_allow_lds(k_big_eval<WANT_H, HL, CT>, lds);
    k_big_eval<WANT_H, HL, CT><<<dim3(b->B), dim3(BT), lds, b->stream>>>(m->dm, b->tmpA, b->tmpB, b->tmpC, eta, dg, dH, b->chart, b->bigws, b->bigws_stride);
}
void launch_big_eval(const rmx_model* m, const rmx_batch* b, bool wantH, double eta, double* dg, double* dH) {
    const bool hl = wantH && big_hl(m), ct = m->dm.con != nullptr;
    const size_t lds = big_dyn_lds(m, hl);
    if (ct) {
        if (hl) big_eval_launch<true, true, true>(m, b, eta, dg, dH, lds);
        else if (wantH) big_eval_launch<true, false, true>(m, b, eta, dg, dH, lds);
        else big_eval_launch<false, false, true>(m, b, eta, dg, dH, lds);
    } else {
        if (hl) big_eval_launch<true, true, false>(m, b, eta, dg, dH, lds);
        else if (wantH) big_eval_launch<true, false, false>(m, b, eta, dg, dH, lds);
        else big_eval_launch<false, false, false>(m, b, eta, dg, dH, lds);
    }
}
void launch_big_energy(const rmx_model* m, const rmx_batch* b, double* dT, double* dV) {
    const dim3 grid(b->B), block(BT);
    const size_t lds = big_dyn_lds(m, false);
    big_allow_lds(k_big_energy, lds);
    k_big_energy<<<grid, block, lds, b->stream>>>(m->dm, b->q, b->qd, dT, dV, b->chart, b->bigws, b->bigws_stride);
}
