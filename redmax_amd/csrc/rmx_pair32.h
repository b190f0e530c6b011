// rmx_pair32.h -- the full 32-link serial chain under BDF1 (BASELINE.json configs[1], the headline workload): every evaluation of the
// front carries TWO points, and the second one is the next step's first.
//
// A tree of <= 32 nodes leaves lanes 32..63 idle in every lane = node stage of the front (1 164 of the 3 014 VALU instructions of a
// Newton iteration of k_step_bdf1<32, false, false, true>).  simLoop (driverRedMaxBDF1.m:57-91) ends a step's solve at the accepted
// line-search trial whose |g| is below tol and starts the next step at  x' = q1 + h qdot1,  q1 = x,  qdot1 = (x - q0) / h  - a point
// that depends on the trial point alone, so it is known BEFORE the trial is evaluated.  eval_front_pair (rmx_ct32.h) evaluates the
// trial (the primary point P) in one half-wave and x' (the secondary point Q) in the other, in one instruction stream.  When P ends
// the solve, the step's epilogue runs, the halves swap roles (`prim`), and the loop goes on as the next step's solve with its first
// evaluation already done: one front per step less (of 5.2), for 46 more LDS / add instructions per front (the 28 subtree sums of two
// points through the accumulation scratch, chain_suffix_sum_pair_lds).  When P does not end the solve, Q is dropped - it cost nothing.
// No prediction is involved: every trial carries its own "what if this is the last one".
//
// ONE loop around ONE call site of the front, the Hessian stage and each solve (newton_rot's lesson: a second producer of the
// front's ~70 doubles costs blocks of register moves on the loop's edges).  Every decision is newton_rot's / newton_policy's /
// k_step_bdf1's, operation for operation; per point the front is eval_front_e2<32, true>'s arithmetic in its order, the Hessian
// stage and the solves are the same functions on the same LDS images: states, iteration and halving counts, status words and
// histories are bit-identical to the one-point kernel (tests/test_gpu_full_size.py::test_chain32_pair_kernel).
// Newton state (x, lo, x0, lo0, dx, q0) is held MIRRORED in both half-waves (lane l and l + 32 hold node l's values), so either half
// can be P; what differs per half is the point handed to the front and the state it leaves (fs, e).
#pragma once

#include "rmx_ct32.h"

namespace rmx {

// sum of v over the lanes of half-wave `half` (wave-uniform), with the bits wave_sum gives for a wave whose other lanes hold zeros
__device__ __forceinline__ double half_sum(const double v, const int half) {
    double sa, sb;
    wave_sum_dual(v, sa, sb);
    return half ? sb : sa;
}

// simLoop: steps 0 .. nsteps - 1 of one rollout.  q, qd: the state of node lane & 31 (mirrored in both half-waves), in and out.
// ENERGY: the call records T, V per step (Scene.saveHistory).  Without it - the benchmark's launch - the energies of the last
// evaluation (`last`, `e0`: six doubles copied at every Newton iteration) and their arithmetic in the front are not carried at all.
template <bool ENERGY>
__device__ __forceinline__ void pair_rollout_bdf1(const DevModel& M, const DevOpts& o, const StepArgs& a, double* sAcc, const int lane,
                                                  const int traj, const int id, const size_t off, double& q, double& qd, int& iters,
                                                  int& halvings, int& status, PivotPolicy& piv) {
    constexpr int NP = 32;
    if (a.nsteps <= 0) return;
    const double* cK = RMX_CONSTS(sAcc, M.n, NP);
    const double grav[3] = {M.grav[0], M.grav[1], M.grav[2]};
#ifndef RMX_PAIR_REGK
#define RMX_PAIR_REGK 1            // 0: the per-node constants from the wave's LDS copy at every evaluation (build variants)
#endif
    double rk[PAIR_NK];            // this lane's per-node constants, in registers for the whole rollout (the kernel holds 270 of 512)
    if constexpr (RMX_PAIR_REGK) pair_load_consts(cK, lane, rk);
    const double h = o.h;
    const bool hiH = lane >= 32;
    FrontState fs;
    NodeOut e, e0, last;
    e0.g = e0.eT = e0.eV = 0.0;
    last = e0;
    int s = 0;
    int prim = 0;                  // the half-wave that carries the primary point
    int lastHalf = 0, e0Half = 0;  // ... and the halves `last` / `e0` were taken in
    double q0 = q;
    double x = fma(h, qd, q0);     // initial guess (:70) and q0 + h qdot0 of dqtmp (:169); (the fused form the one-point kernel compiles to)
    double qB = x;
    double lo = 0.0, dx = 0.0, alpha = 1.0, f0 = 0.0, g0n2 = 0.0, x0 = x, lo0 = 0.0, gn2 = 0.0;
    int iter = 1, lsfail = 0, iterLs = 1;
    bool ls = false;               // the point being evaluated is a trial of the line search (:124-138)
    bool redo = false;             // ... is the re-evaluation before the pivoting re-solve of a tripped guarded solve
    // newton_policy: the linear-solve flavour is chosen per step
    bool pivot_all = o.lu_mode != 0 || piv.hold > 0;
    if (piv.hold > 0) --piv.hold;
    // The end of step s's solve at (x, lo): newton_policy's update, k_step_bdf1's epilogue (qdot :72, q, Scene.saveHistory) and the
    // start of the next solve.  False: that was the last step.
    auto end_step = [&]() -> bool {
        if (!pivot_all) pivot_policy_update(piv);
        qd = ((x - q0) + lo) / h;
        q = x;
        if (ENERGY && a.histT) {
            const double T = half_sum(last.eT, lastHalf), V = half_sum(last.eV, lastHalf);
            if (lane == 0) {
                a.histT[(size_t)s * a.B + traj] = T;
                a.histV[(size_t)s * a.B + traj] = V;
            }
        }
        if (a.histQ && id >= 0 && !hiH) {
            a.histQ[(size_t)s * a.B * M.nr + off] = q;
            a.histQd[(size_t)s * a.B * M.nr + off] = qd;
        }
        ++s;
        if (s >= a.nsteps) return false;
        q0 = q;
        x = fma(h, qd, q0);
        qB = x;
        lo = 0.0; dx = 0.0; alpha = 1.0; f0 = 0.0; g0n2 = 0.0; x0 = x; lo0 = 0.0;
        iter = 1; lsfail = 0; iterLs = 1;
        ls = false;
        redo = false;
        if constexpr (ENERGY) e0.g = e0.eT = e0.eV = 0.0;
        pivot_all = o.lu_mode != 0 || piv.hold > 0;
        if (piv.hold > 0) --piv.hold;
        return true;
    };
    while (true) {
        // ---- the two points of this evaluation.  P = (x, lo) in half `prim`; Q = the first point of step s + 1 if P ends this solve:
        // q' = x, qdot' = ((x - q0) + lo) / h (end_step), x' = q' + h qdot', handed over as newton_rot hands over a first point
        // (x', ((x' - q') + 0) / h, (x' - x') + 0).  redo: P in both halves (the state comes back into half 0).
        const double qdn = ((x - q0) + lo) / h;
        const double xn = fma(h, qdn, x);
        const bool isQ = !redo && ((lane >> 5) != prim);
        const double xe = isQ ? xn : x;
        const double xqd = isQ ? ((xn - x) + 0.0) / h : qdn;
        const double xv = isQ ? ((xn - xn) + 0.0) : ((x - qB) + lo);
        bool ta, tb;
        eval_front_pair<false, true, false, RMX_PAIR_REGK != 0>(M.n, cK, grav, lane, xe, xqd, xv, h, e, fs, ta, tb, sAcc, nullptr, rk);
        double ga2, gb2;
        wave_sum_dual(e.g * e.g, ga2, gb2);
        if (redo) prim = 0;
        else gn2 = prim ? gb2 : ga2;
        if (ls && !redo) {                                   // this was a trial point of the line search (:124-138)
            if (!(0.5 * gn2 < f0) && iterLs < o.iterLsMax) {
                alpha *= 0.5;
                ++iterLs;
                two_sum(x0, fma(alpha, dx, lo0), x, lo);
                lo *= o.comp;
                if (__all(x == x0 && lo == lo0)) {           // see newton_impl: every further halving re-evaluates g(x0)
                    if constexpr (ENERGY) {
                        last = e0;
                        lastHalf = e0Half;
                    }
                    halvings += o.iterLsMax - 1;
                    if (!(sqrt(g0n2) < o.tol)) status |= 2 | 8;
                    if (!end_step()) break;
                }
                continue;
            }
            if constexpr (ENERGY) {
                last = e;
                lastHalf = prim;
            }
            halvings += iterLs - 1;
            bool ends = false;
            if (sqrt(gn2) < o.tol) {
                ends = true;
            } else if (iter >= o.iterMax) {
                status |= 2;
                ends = true;
            } else {
                lsfail += (0.5 * gn2 < f0) ? 0 : 1;
                if (o.lsFailLimit > 0 && lsfail >= o.lsFailLimit) {
                    status |= 2 | ST_LS_CUT;
                    ends = true;
                } else {
                    ++iter;
                }
            }
            if (ends) {
                // the solve of step s ends at P; Q is the first evaluation of step s + 1's solve: the halves swap roles
                const double gq2 = prim ? ga2 : gb2;
                if (!end_step()) break;
                prim ^= 1;
                gn2 = gq2;
            }
        }
        // ---- the Hessian stage on P's state, dx = -H\g
        double Hdummy[NP];
        (void)eval_hess<NP, false, false, false, true>(M, lane, fs, Hdummy, nullptr, sAcc, e.g, prim);
        if constexpr (ENERGY) {
            e0 = e;
            e0Half = prim;
            last = e;
            lastHalf = prim;
        }
        if (!redo) ++iters;
        if (pivot_all || redo) {
            double Hrow[NP];
            hess_rows_from_staging(M.n, lane, sAcc, Hrow);
            const double gl = prim ? take_hi(e.g) : e.g;
            dx = lu_solve_neg<NP>(M.n, lane, Hrow, hiH ? 0.0 : gl);
        } else {
            bool lu_ok;
            dx = lu_solve_neg_diag32(M.n, lane, sAcc, e.g, lu_ok);
            if (!lu_ok) {          // growth guard tripped: redo this solve with partial pivoting (H was eliminated in place)
                ++piv.streak;
                status |= 16;
                redo = true;
                continue;
            }
            piv.streak = 0;
        }
        redo = false;
        dx = dup_lo(dx);           // both solves leave dx in lanes 0..31
        const double dxn2 = wave_sum_np<NP>(dx * dx);
        bool gives_up = false;
        if (!(dxn2 == dxn2)) {
            status |= 4;
            gives_up = true;
        } else if (sqrt(dxn2) > o.dxMax) {
            status |= 1;
            gives_up = true;
        }
        if (gives_up) {            // (:118-121) x is left at the last iterate; stepping continues
            if (!end_step()) break;
            continue;
        }
        alpha = 1.0;
        g0n2 = gn2;
        f0 = 0.5 * g0n2;
        x0 = x;
        lo0 = lo;
        iterLs = 1;
        two_sum(x0, fma(alpha, dx, lo0), x, lo);
        lo *= o.comp;
        if (__all(x == x0 && lo == lo0)) {
            if constexpr (ENERGY) {
                last = e0;
                lastHalf = e0Half;
            }
            halvings += o.iterLsMax - 1;
            if (!(sqrt(g0n2) < o.tol)) status |= 2 | 8;
            if (!end_step()) break;
            continue;
        }
        ls = true;
    }
}

}  // namespace rmx
