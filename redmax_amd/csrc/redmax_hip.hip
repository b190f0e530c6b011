// redmax_hip.hip -- C ABI (include/redmax_hip.h) + kernel entry points for gfx950.
//
// Host side: Scene.init()-equivalent flattening of the scene listing into SoA device constants
// (matlab-diff/+redmax/Scene.m:59-119) and launch plumbing.  Device side: rmx_device.h.
// There is no CPU fallback: every entry point needs a HIP device.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <string>
#include <vector>

#include "redmax_hip.h"
#include "rmx_device.h"

using namespace rmx;

// ============================================================================ kernels

enum { INTEG_BDF1 = 1, INTEG_BDF2 = 2 };

struct StepArgs {
    int B, nsteps;
    double* q;        // [B][nr] state (in/out)
    double* qd;
    double* qp;       // [B][nr] state of step k-1 (BDF2)
    double* qdp;
    int* started;     // [1] device flag: 0 => BDF2 must take the SDIRK2 start step first
    int* it;          // [B] stats (accumulated) or null
    int* ls;
    int* status;
    double* histT;    // [nsteps][B] or null
    double* histV;
};

template <int NP>
__device__ __forceinline__ void smem_setup(const DevModel& M, double*& sAcc, double*& sCol) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    sAcc = smem;
    sCol = smem + (M.n + 1) * ACC_STRIDE;     // per-node constants, [NCONST][NP] (see eval_front_e2)
    if (threadIdx.x < ACC_STRIDE) sAcc[M.n * ACC_STRIDE + threadIdx.x] = 0.0;   // zero row n (end-of-tree suffix)
    if (threadIdx.x < NP) {
        const int j = threadIdx.x;
        const bool in = j < M.n;
        double* c = sCol;
        for (int r = 0; r < 36; ++r) c[r * NP + j] = in ? M.K[r * MAXN + j] : 0.0;
        c += 36 * NP;
        for (int r = 0; r < 6; ++r) c[r * NP + j] = in ? M.sb[r * MAXN + j] : 0.0;
        c += 6 * NP;
        for (int r = 0; r < 4; ++r) c[r * NP + j] = in ? M.I4[r * MAXN + j] : 0.0;
        c += 4 * NP;
        for (int r = 0; r < 8; ++r) c[r * NP + j] = in ? M.prm[r * MAXN + j] : 0.0;
        c += 8 * NP;
        c[j] = in ? (double)M.type[j] : 0.0;
        c += NP;
        c[j] = in ? __longlong_as_double((long long)M.rel[j]) : 0.0;
        c[NP + j] = in ? __longlong_as_double((long long)M.rel[MAXN + j]) : 0.0;
        c += 2 * NP;
        for (int r = 0; r < MAXROUNDS; ++r) c[r * NP + j] = in ? (double)M.anc[r * MAXN + j] : -1.0;
        c += MAXROUNDS * NP;
        c[j] = in ? (double)M.end[j] : (double)M.n;
        c += NP;
        for (int r = 0; r < 4; ++r) c[r * NP + j] = (in && M.con) ? M.con[r * MAXN + j] : 0.0;
    }
    __syncthreads();
}

// simLoop (driverRedMaxBDF1.m:57-91): all steps of one trajectory inside one wavefront.
template <int NP, bool CT>
__global__ void __launch_bounds__(64) k_step_bdf1(const DevModel M, const DevOpts o, const StepArgs a) {
    double *sAcc, *sCol;
    smem_setup<NP>(M, sAcc, sCol);
    const int lane = threadIdx.x, traj = blockIdx.x;
    const int id = (lane < M.n) ? M.idx[lane] : -1;
    const size_t off = (size_t)traj * M.nr + (id >= 0 ? id : 0);
    double q = id >= 0 ? a.q[off] : 0.0;
    double qd = id >= 0 ? a.qd[off] : 0.0;
    int iters = 0, halv = 0, status = 0;
    PivotPolicy piv;
    for (int s = 0; s < a.nsteps; ++s) {
        const double q0 = q, qd0 = qd;
        const double xg = q0 + o.h * qd0;          // initial guess (:70) and q0 + h qdot0 of dqtmp (:169)
        NodeOut last;
        const double x = newton_node<NP, CT>(M, o, sAcc, sCol, lane, xg, q0, xg, o.h, last, iters, halv, status, piv);
        qd = (x - q0) / o.h;                       // (:72)
        q = x;
        if (a.histT) {                             // Scene.saveHistory (Scene.m:134-161)
            const double T = wave_sum(last.eT), V = wave_sum(last.eV);
            if (lane == 0) {
                a.histT[(size_t)s * a.B + traj] = T;
                a.histV[(size_t)s * a.B + traj] = V;
            }
        }
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

// simLoop (driverRedMaxBDF2.m:57-125): SDIRK2 start step (two Newton solves), then BDF2.
template <int NP, bool CT>
__global__ void __launch_bounds__(64) k_step_bdf2(const DevModel M, const DevOpts o, const StepArgs a) {
    double *sAcc, *sCol;
    smem_setup<NP>(M, sAcc, sCol);
    const int lane = threadIdx.x, traj = blockIdx.x;
    const int id = (lane < M.n) ? M.idx[lane] : -1;
    const size_t off = (size_t)traj * M.nr + (id >= 0 ? id : 0);
    double q = id >= 0 ? a.q[off] : 0.0;
    double qd = id >= 0 ? a.qd[off] : 0.0;
    double qp = id >= 0 ? a.qp[off] : 0.0;       // step k-1 (Joint.q1 / qdot1 in the reference)
    double qdp = id >= 0 ? a.qdp[off] : 0.0;
    const bool started = (*a.started) != 0;
    const double h = o.h;
    int iters = 0, halv = 0, status = 0;
    PivotPolicy piv;
    for (int s = 0; s < a.nsteps; ++s) {
        NodeOut last;
        if (s == 0 && !started) {
            const double al = (2.0 - sqrt(2.0)) / 2.0;    // (:74)
            const double q0 = q, qd0 = qd;
            // SDIRK2a (evalSDIRK2a :194-225): eta = a h, qA = q0, qB = q0 + a h qdot0
            const double xa0 = q0 + al * h * qd0;
            const double qa = newton_node<NP, CT>(M, o, sAcc, sCol, lane, xa0, q0, q0 + (al * h) * qd0, al * h, last, iters, halv, status, piv);
            const double qda = (qa - q0) / (al * h);
            // SDIRK2b (evalSDIRK2b :228-260)
            const double x10 = qa + (1.0 - al) * h * qda;
            const double qA = q0 + (1.0 - al) * h * qda;
            const double qB = q0 + (2.0 * al - 1.0) * h * qd0 + 2.0 * (1.0 - al) * h * qda;
            const double q1 = newton_node<NP, CT>(M, o, sAcc, sCol, lane, x10, qA, qB, al * h, last, iters, halv, status, piv);
            qd = (q1 - q0 - (1.0 - al) * h * qda) / (al * h);
            q = q1;
            qp = q0;
            qdp = qd0;
        } else {
            // BDF2 (evalBDF2 :263-293): eta = 2h/3
            const double q0 = qp, qd0 = qdp, q1 = q, qd1 = qd;
            const double x0 = q1 + h * qd1;
            const double qA = (4.0 / 3.0) * q1 - (1.0 / 3.0) * q0;
            const double qB = (4.0 / 3.0) * q1 - (1.0 / 3.0) * q0 + (8.0 / 9.0) * h * qd1 - (2.0 / 9.0) * h * qd0;
            const double q2 = newton_node<NP, CT>(M, o, sAcc, sCol, lane, x0, qA, qB, (2.0 / 3.0) * h, last, iters, halv, status, piv);
            qp = q1;
            qdp = qd1;
            qd = (3.0 / (2.0 * h)) * (q2 - (4.0 / 3.0) * q1 + (1.0 / 3.0) * q0);
            q = q2;
            // the Newton residual was evaluated with qdot = (q2-qA)/eta, identical up to rounding
        }
        if (a.histT) {
            const double T = wave_sum(last.eT), V = wave_sum(last.eV);
            if (lane == 0) {
                a.histT[(size_t)s * a.B + traj] = T;
                a.histV[(size_t)s * a.B + traj] = V;
            }
        }
    }
    if (id >= 0) {
        a.q[off] = q;
        a.qd[off] = qd;
        a.qp[off] = qp;
        a.qdp[off] = qdp;
    }
    if (lane == 0 && a.it) {
        a.it[traj] += iters;
        a.ls[traj] += halv;
        a.status[traj] |= status;
    }
}

// euler (matlab-simple/testRedMax.m:67-109), BASELINE.json configs[0]: linearly-implicit Euler,
//   Mr = J'MmJ ; (Mr + h Dr - h^2 Kr) qdot1 = Mr qdot0 + h (J'(fm - Mm Jdot qdot0) + fr) ; q1 = q0 + h qdot1
// with the same front as the implicit integrators: the right-hand side is g(v = qdot0, e2 = -h) plus h*damping*qdot0
// (matlab-simple drops the joint damping FORCE, testRedMax.m:84), the matrix is eval_mass + the joint diagonals.
template <int NP>
__global__ void __launch_bounds__(64) k_step_euler(const DevModel M, const double h, const StepArgs a) {
    double *sAcc, *sCol;
    smem_setup<NP>(M, sAcc, sCol);
    const int lane = threadIdx.x, traj = blockIdx.x;
    const int id = (lane < M.n) ? M.idx[lane] : -1;
    const size_t off = (size_t)traj * M.nr + (id >= 0 ? id : 0);
    double q = id >= 0 ? a.q[off] : 0.0;
    double qd = id >= 0 ? a.qd[off] : 0.0;
    for (int s = 0; s < a.nsteps; ++s) {
        FrontState fs;
        NodeOut e;
        double Mrow[NP];
        eval_front_e2<NP, true>(M, sAcc, lane, q, qd, qd, 0.0, -h, e, fs);
        eval_mass<NP>(M, lane, fs, Mrow);
        const double rhs = e.g + h * fs.dd * qd;
#pragma unroll
        for (int i = 0; i < NP; ++i)
            if (i == lane && fs.dof) Mrow[i] += h * fs.dd + h * h * fs.kd;
        const double qd1 = lu_solve_neg<NP>(M.n, lane, Mrow, -rhs);
        q = q + h * qd1;
        qd = qd1;
        if (a.histT) {
            eval_front<NP, false>(M, sAcc, lane, q, qd, 0.0, 1.0, e, fs);
            const double T = wave_sum(e.eT), V = wave_sum(e.eV);
            if (lane == 0) {
                a.histT[(size_t)s * a.B + traj] = T;
                a.histV[(size_t)s * a.B + traj] = V;
            }
        }
    }
    if (id >= 0) {
        a.q[off] = q;
        a.qd[off] = qd;
    }
}

// ---------------------------------------------------------------- adjoint BDF1 (BASELINE.json configs[3])
//
// taskObjective of driverRedMaxAdjointBDF1.m:39-62 for TaskBDF1PointPos, batched: parameters p[B][nr] are constant joint
// torques tau = pscale*p (TaskBDF1PointPos.applyStep :58-64).  Forward = simLoop :65-102 with the line-search-free newton
// :105-146; per step the H, M, D of the LAST EVALUATED iterate are kept in HBM ([B][nsteps][n*n], column-major over nodes),
// which is what Scene.saveHistory stores (the reference keeps lu(H), the backward kernel re-factors H').  Backward =
// TaskBDF1.calcFinal :45-81.
struct AdjArgs {
    int B, nsteps, task_step, task_node;
    double xl[3], xt[3], pscale, wreg, wpos;
    double *q, *qd;
    const double* p;
    double *Hs, *Ms, *Ds;     // [B][nsteps][n*n]
    double* dPdq;             // [B][n]   dP/dq of the task step
    double* P;                // [B]
    double* dPdp;             // [B][nr]
    int *it, *status;
};

template <int NP>
__global__ void __launch_bounds__(64) k_adjoint_fwd(const DevModel M, const DevOpts o, const AdjArgs a) {
    double *sAcc, *sCol;
    smem_setup<NP>(M, sAcc, sCol);
    const int lane = threadIdx.x, traj = blockIdx.x, n = M.n;
    const int id = (lane < n) ? M.idx[lane] : -1;
    const size_t off = (size_t)traj * M.nr + (id >= 0 ? id : 0);
    double q = id >= 0 ? a.q[off] : 0.0;
    double qd = id >= 0 ? a.qd[off] : 0.0;
    const double pj = id >= 0 ? a.p[off] : 0.0;
    const double h = o.h;
    FrontState fs;
    fs.tau_add = a.pscale * pj;
    // does the task body hang below (or at) this lane's joint?  (rows of J(idxM_body, :) that are non-zero)
    const bool on_path = lane < n && (lane == a.task_node || ((M.rel[MAXN + lane] >> a.task_node) & 1ull));
    int iters = 0, status = 0;
    double Ptask = 0.0;
    const size_t nn = (size_t)n * n;
    for (int s = 1; s <= a.nsteps; ++s) {
        const double q0 = q, qd0 = qd;
        const double xB = q0 + h * qd0;
        double x = xB;
        double* Hk = a.Hs + ((size_t)traj * a.nsteps + (s - 1)) * nn;
        double* Mk = a.Ms + ((size_t)traj * a.nsteps + (s - 1)) * nn;
        double* Dk = a.Ds + ((size_t)traj * a.nsteps + (s - 1)) * nn;
        double Jw[3] = {0.0, 0.0, 0.0}, Jv[3] = {0.0, 0.0, 0.0};   // J(idxM_body, this joint) of the last evaluated iterate
        int iter = 1;
        while (true) {
            NodeOut e;
            double Hrow[NP];
            eval_front<NP, true>(M, sAcc, lane, x, (x - q0) / h, x - xB, h, e, fs);
            eval_hess<NP>(M, lane, fs, Hrow);
            {
                double Mrow[NP], Drow[NP];
                eval_MD<NP>(M, lane, fs, Mrow, Drow);
                if (lane < n) {
#pragma unroll
                    for (int i = 0; i < NP; ++i)
                        if (i < n) {
                            Hk[(size_t)i * n + lane] = Hrow[i];
                            Mk[(size_t)i * n + lane] = Mrow[i];
                            Dk[(size_t)i * n + lane] = Drow[i];
                        }
                }
            }
            if (s == a.task_step) {   // J(body rows, joint) = Ad(E_body^-1) s_joint : body-frame twist of the task body per unit qdot
                double Rb[9], pb[3], t3[3], d3[3];
#pragma unroll
                for (int c = 0; c < 9; ++c) Rb[c] = readlane_d(fs.Rw[c], a.task_node);
#pragma unroll
                for (int c = 0; c < 3; ++c) pb[c] = readlane_d(fs.pw[c], a.task_node);
                cross3(pb, fs.sw, t3);
#pragma unroll
                for (int c = 0; c < 3; ++c) d3[c] = fs.sv[c] - t3[c];
#pragma unroll
                for (int c = 0; c < 3; ++c) {   // R' (.)
                    Jw[c] = on_path ? (Rb[c] * fs.sw[0] + Rb[3 + c] * fs.sw[1] + Rb[6 + c] * fs.sw[2]) : 0.0;
                    Jv[c] = on_path ? (Rb[c] * d3[0] + Rb[3 + c] * d3[1] + Rb[6 + c] * d3[2]) : 0.0;
                }
            }
            ++iters;
            const double dx = lu_solve_neg<NP>(n, lane, Hrow, e.g);      // [Hl,Hu,Hp] = lu(H,'vector'); dx = -(Hu\(Hl\g(Hp)))  :127-128
            const double dxn2 = wave_sum(dx * dx);
            if (!(dxn2 == dxn2)) { status |= 4; break; }
            if (sqrt(dxn2) > o.dxMax) { status |= 1; break; }            // :129-132
            x = x + dx;                                                   // :134, before the convergence test
            if (sqrt(wave_sum(e.g * e.g)) < o.tol) break;                 // :135-138
            if (iter >= o.iterMax) { status |= 2; break; }                // :139-142
            ++iter;
        }
        qd = (x - q0) / h;
        q = x;
        if (s == a.task_step) {    // TaskBDF1PointPos.calcStep :67-107 at the final state of this step
            NodeOut e;
            eval_front<NP, false>(M, sAcc, lane, q, qd, 0.0, 1.0, e, fs);
            double Rb[9], pb[3], dxw[3], vl[3], t3[3];
#pragma unroll
            for (int c = 0; c < 9; ++c) Rb[c] = readlane_d(fs.Rw[c], a.task_node);
#pragma unroll
            for (int c = 0; c < 3; ++c) pb[c] = readlane_d(fs.pw[c], a.task_node);
#pragma unroll
            for (int c = 0; c < 3; ++c) dxw[c] = Rb[3 * c] * a.xl[0] + Rb[3 * c + 1] * a.xl[1] + Rb[3 * c + 2] * a.xl[2] + pb[c] - a.xt[c];
            Ptask += a.wpos * 0.5 * dot3(dxw, dxw);
            // dPdq = J' * (R*Gamma(xlocal))' * dx * wp ,  Gamma = [brac(xlocal)', I]  =>  R (v + w x xlocal) . dx * wp
            const double xl[3] = {a.xl[0], a.xl[1], a.xl[2]};
            cross3(Jw, xl, t3);
#pragma unroll
            for (int c = 0; c < 3; ++c) vl[c] = Jv[c] + t3[c];
            double g3[3];
            mat3v(Rb, vl, g3);
            if (lane < n) a.dPdq[(size_t)traj * n + lane] = a.wpos * dot3(g3, dxw);
        }
    }
    if (id >= 0) {
        a.q[off] = q;
        a.qd[off] = qd;
    }
    const double preg = wave_sum(pj * pj);
    if (lane == 0) {
        a.P[traj] = Ptask + a.wreg * 0.5 * preg;      // TaskBDF1.calcFinal :49
        if (a.it) {
            a.it[traj] = iters;
            a.status[traj] = status;
        }
    }
}

template <int NP>
__global__ void __launch_bounds__(64) k_adjoint_bwd(const DevModel M, const DevOpts o, const AdjArgs a) {
    const int lane = threadIdx.x, traj = blockIdx.x, n = M.n;
    const int id = (lane < n) ? M.idx[lane] : -1;
    const size_t nn = (size_t)n * n;
    const double h = o.h;
    const int col = lane < n ? lane : 0;
    double z1 = 0.0, z2 = 0.0, zs = 0.0;
    for (int k = a.nsteps; k >= 1; --k) {
        double y = (k == a.task_step && lane < n) ? a.dPdq[(size_t)traj * n + lane] : 0.0;
        if (k < a.nsteps) {       // yk -= (-2 M_{k+1} + h D_{k+1})' z_{k+1}     TaskBDF1.m:58-64
            const double* Mc = a.Ms + ((size_t)traj * a.nsteps + k) * nn + (size_t)col * n;
            const double* Dc = a.Ds + ((size_t)traj * a.nsteps + k) * nn + (size_t)col * n;
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const double blk = (j < n && lane < n) ? (-2.0 * Mc[j] + h * Dc[j]) : 0.0;
                y -= blk * readlane_d(z1, j);
            }
        }
        if (k < a.nsteps - 1) {   // yk -= M_{k+2}' z_{k+2}                      :65-70
            const double* Mc = a.Ms + ((size_t)traj * a.nsteps + k + 1) * nn + (size_t)col * n;
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const double blk = (j < n && lane < n) ? Mc[j] : 0.0;
                y -= blk * readlane_d(z2, j);
            }
        }
        // z_k = H_k'^-1 y_k  (zkk0(Hp) = Hl'\(Hu'\yk) :76): this lane's "row" of H' is column `lane` of H
        double Hrow[NP];
        const double* Hc = a.Hs + ((size_t)traj * a.nsteps + (k - 1)) * nn + (size_t)col * n;
#pragma unroll
        for (int i = 0; i < NP; ++i) Hrow[i] = (i < n && lane < n) ? Hc[i] : ((i == lane) ? 1.0 : 0.0);
        const double z = lu_solve_neg<NP>(n, lane, Hrow, -y);
        zs += z;
        z2 = z1;
        z1 = z;
    }
    if (id >= 0) {   // dPdp = wreg*p' - z'*dgdp, dgdp(kk,:) = -h^2*pscale*I      :79, TaskBDF1PointPos.m:104-105
        const size_t off = (size_t)traj * M.nr + id;
        a.dPdp[off] = a.wreg * a.p[off] + h * h * a.pscale * zs;
    }
}

// Parity hook: one residual (+Hessian) evaluation per trajectory, results to HBM.
template <int NP, bool WANT_H, bool CT>
__global__ void __launch_bounds__(64) k_eval(const DevModel M, const int B, const double* __restrict__ q,
                                             const double* __restrict__ qA, const double* __restrict__ qB, const double eta,
                                             double* __restrict__ g, double* __restrict__ H) {
    double *sAcc, *sCol;
    smem_setup<NP>(M, sAcc, sCol);
    const int lane = threadIdx.x, traj = blockIdx.x;
    const int id = (lane < M.n) ? M.idx[lane] : -1;
    const size_t off = (size_t)traj * M.nr + (id >= 0 ? id : 0);
    const double x = id >= 0 ? q[off] : 0.0;
    const double xa = id >= 0 ? qA[off] : 0.0;
    const double xb = id >= 0 ? qB[off] : 0.0;
    NodeOut e;
    double Hrow[NP];
    eval_node<NP, WANT_H, false, CT>(M, sAcc, sCol, lane, x, (x - xa) / eta, x - xb, eta, e, Hrow);
    if (id >= 0) g[off] = e.g;
    if (WANT_H) {
        double* Ht = H + (size_t)traj * M.nr * M.nr;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            if (i < M.n) {
                const int ci = M.idx[i];
                if (id >= 0 && ci >= 0) Ht[(size_t)ci * M.nr + id] = Hrow[i];   // column-major H(id, ci)
            }
        }
    }
}

// Joint.computeEnergies / Body.computeEnergies at the stored state.
template <int NP, bool CT>
__global__ void __launch_bounds__(64) k_energy(const DevModel M, const int B, const double* __restrict__ q,
                                               const double* __restrict__ qd, double* __restrict__ T, double* __restrict__ V) {
    double *sAcc, *sCol;
    smem_setup<NP>(M, sAcc, sCol);
    const int lane = threadIdx.x, traj = blockIdx.x;
    const int id = (lane < M.n) ? M.idx[lane] : -1;
    const size_t off = (size_t)traj * M.nr + (id >= 0 ? id : 0);
    NodeOut e;
    double Hrow[NP];
    eval_node<NP, false, false, CT>(M, sAcc, sCol, lane, id >= 0 ? q[off] : 0.0, id >= 0 ? qd[off] : 0.0, 0.0, 1.0, e, Hrow);
    const double t = wave_sum(e.eT), v = wave_sum(e.eV);
    if (lane == 0) {
        T[traj] = t;
        V[traj] = v;
    }
}

// Profiling hook: shader-clock cycles (s_memtime) of the phases of one Newton iteration, measured in place with the
// production device functions at the production occupancy (one wavefront per trajectory).
template <int NP>
__global__ void __launch_bounds__(64) k_phase_time(const DevModel M, const int reps, const double* __restrict__ q,
                                                   const double* __restrict__ qd, const double h, unsigned long long* __restrict__ out) {
    double *sAcc, *sCol;
    smem_setup<NP>(M, sAcc, sCol);
    const int lane = threadIdx.x, traj = blockIdx.x;
    const int id = (lane < M.n) ? M.idx[lane] : -1;
    const size_t off = (size_t)traj * M.nr + (id >= 0 ? id : 0);
    const double q0 = id >= 0 ? q[off] : 0.0, qd0 = id >= 0 ? qd[off] : 0.0;
    double x = q0 + h * qd0;
    unsigned long long tg = 0, tH = 0, tLU = 0, tred = 0;
    unsigned long long stamps[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    double sink = 0.0;
    for (int r = 0; r < reps; ++r) {
        NodeOut e;
        double Hrow[NP];
        unsigned long long t0 = __builtin_amdgcn_s_memtime();
        eval_node<NP, false>(M, sAcc, sCol, lane, x, (x - q0) / h, x - (q0 + h * qd0), h, e, Hrow);
        sink += e.g;
        unsigned long long t1 = __builtin_amdgcn_s_memtime();
        eval_node<NP, true, true>(M, sAcc, sCol, lane, x, (x - q0) / h, x - (q0 + h * qd0), h, e, Hrow, stamps);
        unsigned long long t2 = __builtin_amdgcn_s_memtime();
        const double dx = lu_solve_neg<NP>(M.n, lane, Hrow, e.g);
        unsigned long long t3 = __builtin_amdgcn_s_memtime();
        const double s1 = wave_sum(dx * dx) + wave_sum(e.g * e.g);
        unsigned long long t4 = __builtin_amdgcn_s_memtime();
        sink += s1;
        x += 1e-3 * dx;   // keep the iterations data dependent
        tg += t1 - t0; tH += t2 - t1; tLU += t3 - t2; tred += t4 - t3;
    }
    if (lane == 0) {
        out[16 * traj + 0] = tg; out[16 * traj + 1] = tH; out[16 * traj + 2] = tLU; out[16 * traj + 3] = tred;
        for (int k = 0; k < 12; ++k) out[16 * traj + 4 + k] = stamps[k];
    }
    if (sink == 1.2345e301) out[0] = 0;   // keep the results live
}

// ============================================================================ host side

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
#define HIPCHK(expr)                                                                                          \
    do {                                                                                                      \
        hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess) return fail(RMX_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));      \
    } while (0)

struct rmx_model {
    int device = 0;
    int n = 0, nr = 0, nm = 0, NP = 0;
    std::vector<int> idx_listing;   // reduced index per LISTED joint (-1 fixed)
    std::vector<int> node_of_listing;   // depth-first node index of each LISTED joint/body
    void* dbuf = nullptr;           // one device allocation holding all constant arrays
    void* dcon = nullptr;           // contact flags + cuboid sides (rmx_model_set_ground_contact)
    DevModel dm{};
    size_t smem_bytes = 0;
};

struct rmx_batch {
    rmx_model* m = nullptr;
    int B = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double *q = nullptr, *qd = nullptr, *qp = nullptr, *qdp = nullptr;
    double *tmpA = nullptr, *tmpB = nullptr, *tmpC = nullptr;   // [B][nr] scratch for rmx_eval inputs
    int* started = nullptr;
    int *it = nullptr, *ls = nullptr, *status = nullptr;
    double last_ms = 0.0;
};

extern "C" const char* rmx_last_error(void) { return g_err.c_str(); }
extern "C" int rmx_version(void) { return RMX_VERSION; }
extern "C" int rmx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
extern "C" void rmx_opts_default(rmx_opts* o) {
    if (!o) return;
    o->h = 1e-2;            // Scene.h default (Scene.m:42)
    o->tol = 1e-9;          // driverRedMaxBDF1.m:95
    o->dxMax = 1e3;         // :96
    o->iterMaxPerDof = 10;  // :97
    o->iterLsMax = 20;      // :98
    o->lu_mode = 0;
}

namespace {

struct M4 {
    double a[4][4];
};
M4 eye4() {
    M4 E{};
    for (int i = 0; i < 4; ++i) E.a[i][i] = 1.0;
    return E;
}
M4 from_cm(const double* cm) {
    M4 E;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) E.a[i][j] = cm[j * 4 + i];
    return E;
}
M4 mul(const M4& A, const M4& B) {
    M4 C{};
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0;
            for (int k = 0; k < 4; ++k) s += A.a[i][k] * B.a[k][j];
            C.a[i][j] = s;
        }
    return C;
}
M4 inv(const M4& E) {   // se3.inv (se3.m:11-16)
    M4 I = eye4();
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) I.a[i][j] = E.a[j][i];
    for (int i = 0; i < 3; ++i) {
        double s = 0;
        for (int k = 0; k < 3; ++k) s += E.a[k][i] * E.a[k][3];
        I.a[i][3] = -s;
    }
    return I;
}

// 3x3 helpers on row-major arrays
void m3mul(const double A[9], const double B[9], double C[9]) {
    double T[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) T[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
    std::memcpy(C, T, sizeof(T));
}
void m3v(const double A[9], const double x[3], double y[3]) {
    double t[3];
    for (int i = 0; i < 3; ++i) t[i] = A[3 * i] * x[0] + A[3 * i + 1] * x[1] + A[3 * i + 2] * x[2];
    std::memcpy(y, t, sizeof(t));
}

}  // namespace

extern "C" int rmx_model_create(const rmx_model_desc* d, int device, rmx_model** out) {
    if (!d || !out) return fail(RMX_E_INVALID, "null argument");
    *out = nullptr;
    int ndev = rmx_device_count();
    if (ndev <= 0) return fail(RMX_E_NODEVICE, "no HIP device visible: redmax_hip has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(RMX_E_INVALID, "device index out of range");
    const int n = d->njoints;
    if (n < 1 || n > MAXN) return fail(RMX_E_INVALID, "njoints must be in [1," + std::to_string(MAXN) + "] (one wavefront per tree)");
    if (!d->parent || !d->type || !d->axis || !d->E0_pj || !d->E0_ji || !d->I_i)
        return fail(RMX_E_INVALID, "parent/type/axis/E0_pj/E0_ji/I_i are required");
    // ---- validate the listing: exactly one root first, parents before children (Scene.m:66-67)
    for (int i = 0; i < n; ++i) {
        if (d->type[i] < 0 || d->type[i] > 2) return fail(RMX_E_INVALID, "unsupported joint type (only fixed/revolute/prismatic are in scope)");
        if (i == 0 && d->parent[i] != -1) return fail(RMX_E_INVALID, "joint 0 must be the root");
        if (i > 0 && (d->parent[i] < 0 || d->parent[i] >= i)) return fail(RMX_E_INVALID, "joints must be listed parent-before-child with a single root");
    }
    // ---- reduced / maximal numbering, leaf-to-root over the LISTING (Scene.m:69-71, Joint.countDofs, Body.countDofs)
    std::vector<int> idxL(n, -1);
    int nr = 0;
    for (int i = n - 1; i >= 0; --i)
        if (d->type[i] != RMX_JOINT_FIXED) idxL[i] = nr++;
    // ---- depth-first order (children in listing order); identity for the reference's scenes
    std::vector<std::vector<int>> kids(n);
    for (int i = 1; i < n; ++i) kids[d->parent[i]].push_back(i);
    std::vector<int> order;   // order[k] = listing index of the k-th node in depth-first order
    order.reserve(n);
    {
        std::vector<int> stack{0};
        while (!stack.empty()) {
            int j = stack.back();
            stack.pop_back();
            order.push_back(j);
            for (int c = (int)kids[j].size() - 1; c >= 0; --c) stack.push_back(kids[j][c]);
        }
    }
    std::vector<int> pos(n);
    for (int k = 0; k < n; ++k) pos[order[k]] = k;
    std::vector<int> par(n), endv(n), depth(n);
    for (int k = 0; k < n; ++k) {
        int pl = d->parent[order[k]];
        par[k] = pl < 0 ? -1 : pos[pl];
        depth[k] = pl < 0 ? 0 : depth[par[k]] + 1;
    }
    for (int k = n - 1; k >= 0; --k) {
        endv[k] = k + 1;
        for (int c : kids[order[k]]) endv[k] = std::max(endv[k], endv[pos[c]]);
    }
    int maxdepth = 0, is_chain = 1;
    for (int k = 0; k < n; ++k) {
        maxdepth = std::max(maxdepth, depth[k]);
        if (endv[k] != n) is_chain = 0;
    }
    int rounds = 0;
    while ((1 << rounds) < maxdepth + 1) ++rounds;

    // ---- constants per node
    std::vector<double> K(36 * MAXN, 0.0), sb(6 * MAXN, 0.0), I4(4 * MAXN, 0.0), prm(8 * MAXN, 0.0);
    std::vector<int> type(MAXN, 0), idx(MAXN, -1), endd(MAXN, 0), anc(MAXROUNDS * MAXN, -1);
    std::vector<unsigned long long> rel(2 * MAXN, 0ull);
    for (int k = 0; k < n; ++k) {
        const int L = order[k];
        type[k] = d->type[L];
        idx[k] = idxL[L];
        endd[k] = endv[k];
        int a = par[k];
        // ancestor 2^r levels up
        {
            std::vector<int> chain;   // chain[t] = ancestor t+1 levels up
            for (int t = par[k]; t >= 0; t = par[t]) {
                chain.push_back(t);
                rel[k] |= 1ull << t;            // t is a strict ancestor of k
                rel[MAXN + t] |= 1ull << k;     // k is a strict descendant of t
            }
            for (int r = 0; r < MAXROUNDS; ++r) {
                int lv = 1 << r;
                anc[r * MAXN + k] = (lv <= (int)chain.size()) ? chain[lv - 1] : -1;
            }
        }
        (void)a;
        const M4 E0_pj = from_cm(d->E0_pj + 16 * L);
        const M4 E0_ji = from_cm(d->E0_ji + 16 * L);
        M4 Lm = E0_pj;   // root: E_wj = E0_pj Q  (Joint.m:404-415)
        if (d->parent[L] >= 0) Lm = mul(inv(from_cm(d->E0_ji + 16 * d->parent[L])), E0_pj);   // parent body -> joint frame
        double ax[3] = {d->axis[3 * L], d->axis[3 * L + 1], d->axis[3 * L + 2]};
        if (type[k] != RMX_JOINT_FIXED) {   // JointRevolute.m:14 / JointPrismatic.m:15
            double nn = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
            if (!(nn > 0)) return fail(RMX_E_INVALID, "zero joint axis");
            for (double& v : ax) v /= nn;
        }
        double LR[9], Lp[3], RR[9], Rp[3];
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) {
                LR[3 * i + j] = Lm.a[i][j];
                RR[3 * i + j] = E0_ji.a[i][j];
            }
            Lp[i] = Lm.a[i][3];
            Rp[i] = E0_ji.a[i][3];
        }
        double K0R[9], K0p[3], K1R[9] = {0}, K1p[3] = {0}, K2R[9] = {0}, K2p[3] = {0};
        if (type[k] == RMX_JOINT_REVOLUTE) {
            // se3.aaToMat (se3.m:111-176) special-cases axis-aligned rotations: snap the ROTATION axis exactly
            // as those branches do, so R(q) has the same exact zeros / ones; S keeps the given axis.
            double ar[3] = {ax[0], ax[1], ax[2]};
            const double TH = 1e-9;
            if (std::fabs(ar[0]) < TH && std::fabs(ar[1]) < TH) { ar[0] = 0; ar[1] = 0; ar[2] = ar[2] < 0 ? -1.0 : 1.0; }
            else if (std::fabs(ar[1]) < TH && std::fabs(ar[2]) < TH) { ar[1] = 0; ar[2] = 0; ar[0] = ar[0] < 0 ? -1.0 : 1.0; }
            else if (std::fabs(ar[2]) < TH && std::fabs(ar[0]) < TH) { ar[2] = 0; ar[0] = 0; ar[1] = ar[1] < 0 ? -1.0 : 1.0; }
            // R(q) = a a' + cos q (I - a a') + sin q [a]
            double aat[9], Ima[9], ab[9] = {0, -ar[2], ar[1], ar[2], 0, -ar[0], -ar[1], ar[0], 0};
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) {
                    aat[3 * i + j] = ar[i] * ar[j];
                    Ima[3 * i + j] = (i == j ? 1.0 : 0.0) - ar[i] * ar[j];
                }
            double T[9], t3[3];
            m3mul(LR, aat, T); m3mul(T, RR, K0R); m3v(T, Rp, t3);
            for (int i = 0; i < 3; ++i) K0p[i] = Lp[i] + t3[i];
            m3mul(LR, ab, T); m3mul(T, RR, K1R); m3v(T, Rp, K1p);
            m3mul(LR, Ima, T); m3mul(T, RR, K2R); m3v(T, Rp, K2p);
        } else {
            double t3[3];
            m3mul(LR, RR, K0R);
            m3v(LR, Rp, t3);
            for (int i = 0; i < 3; ++i) K0p[i] = Lp[i] + t3[i];
            if (type[k] == RMX_JOINT_PRISMATIC) m3v(LR, ax, K1p);   // p(q) = a q  (JointPrismatic.m:29-33)
        }
        for (int c = 0; c < 9; ++c) {
            K[c * MAXN + k] = K0R[c];
            K[(12 + c) * MAXN + k] = K1R[c];
            K[(24 + c) * MAXN + k] = K2R[c];
        }
        for (int c = 0; c < 3; ++c) {
            K[(9 + c) * MAXN + k] = K0p[c];
            K[(21 + c) * MAXN + k] = K1p[c];
            K[(33 + c) * MAXN + k] = K2p[c];
        }
        // body-frame joint screw A0_ij S  (Body.setBodyTransform Body.m:46-51, Joint.m:508): Ad(E0_ij) [w; v]
        {
            const M4 E0_ij = inv(E0_ji);
            double S[6] = {0, 0, 0, 0, 0, 0};
            if (type[k] == RMX_JOINT_REVOLUTE) { S[0] = ax[0]; S[1] = ax[1]; S[2] = ax[2]; }
            if (type[k] == RMX_JOINT_PRISMATIC) { S[3] = ax[0]; S[4] = ax[1]; S[5] = ax[2]; }
            double Rm[9], pm[3] = {E0_ij.a[0][3], E0_ij.a[1][3], E0_ij.a[2][3]};
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) Rm[3 * i + j] = E0_ij.a[i][j];
            double w3[3], v3[3];
            m3v(Rm, S, w3);
            m3v(Rm, S + 3, v3);
            const double cx[3] = {pm[1] * w3[2] - pm[2] * w3[1], pm[2] * w3[0] - pm[0] * w3[2], pm[0] * w3[1] - pm[1] * w3[0]};
            for (int c = 0; c < 3; ++c) {
                sb[c * MAXN + k] = w3[c];
                sb[(3 + c) * MAXN + k] = v3[c] + cx[c];
            }
        }
        // inertia: the reference allows a general diagonal, but mass entries must agree (Body.m:107 uses M_i(4,4))
        const double* Ii = d->I_i + 6 * L;
        I4[0 * MAXN + k] = Ii[0];
        I4[1 * MAXN + k] = Ii[1];
        I4[2 * MAXN + k] = Ii[2];
        I4[3 * MAXN + k] = Ii[3];
        if (Ii[3] != Ii[4] || Ii[3] != Ii[5]) return fail(RMX_E_INVALID, "I_i(4:6) must all equal the body mass");
        prm[0 * MAXN + k] = d->tau ? d->tau[L] : 0.0;
        prm[1 * MAXN + k] = d->stiffness ? d->stiffness[L] : 0.0;
        prm[2 * MAXN + k] = d->damping ? d->damping[L] : 0.0;
        prm[3 * MAXN + k] = d->qRest ? d->qRest[L] : 0.0;
        prm[4 * MAXN + k] = d->qLimL ? d->qLimL[L] : -1e8;   // Joint.m:77-80 defaults
        prm[5 * MAXN + k] = d->qLimU ? d->qLimU[L] : 1e8;
        prm[6 * MAXN + k] = d->qLimK ? d->qLimK[L] : 1e8;
        prm[7 * MAXN + k] = d->qLimD ? d->qLimD[L] : 0.0;
    }

    rmx_model* m = new rmx_model();
    m->device = device;
    m->n = n;
    m->nr = nr;
    m->nm = 6 * n;
    m->idx_listing = idxL;
    m->node_of_listing = pos;
    m->NP = n <= 4 ? 4 : n <= 8 ? 8 : n <= 16 ? 16 : n <= 32 ? 32 : 64;
    m->smem_bytes = sizeof(double) * ((size_t)(n + 1) * ACC_STRIDE + (size_t)NCONST * m->NP);
    if (hipSetDevice(device) != hipSuccess) { delete m; return fail(RMX_E_HIP, "hipSetDevice failed"); }
    const size_t nd = K.size() + sb.size() + I4.size() + prm.size();
    const size_t ni = type.size() + idx.size() + endd.size() + anc.size();
    const size_t bytes = nd * sizeof(double) + rel.size() * sizeof(unsigned long long) + ni * sizeof(int);
    hipError_t e = hipMalloc(&m->dbuf, bytes);
    if (e != hipSuccess) { delete m; return fail(RMX_E_NOMEM, std::string("hipMalloc(model): ") + hipGetErrorString(e)); }
    std::vector<char> host(bytes);
    char* hp = host.data();
    char* dp = (char*)m->dbuf;
    auto put = [&](const void* src, size_t nb) { std::memcpy(hp, src, nb); const void* dptr = dp; hp += nb; dp += nb; return dptr; };
    m->dm.n = n;
    m->dm.nr = nr;
    m->dm.rounds = rounds;
    m->dm.is_chain = is_chain;
    m->dm.K = (const double*)put(K.data(), K.size() * sizeof(double));
    m->dm.sb = (const double*)put(sb.data(), sb.size() * sizeof(double));
    m->dm.I4 = (const double*)put(I4.data(), I4.size() * sizeof(double));
    m->dm.prm = (const double*)put(prm.data(), prm.size() * sizeof(double));
    m->dm.rel = (const unsigned long long*)put(rel.data(), rel.size() * sizeof(unsigned long long));
    m->dm.type = (const int*)put(type.data(), type.size() * sizeof(int));
    m->dm.idx = (const int*)put(idx.data(), idx.size() * sizeof(int));
    m->dm.end = (const int*)put(endd.data(), endd.size() * sizeof(int));
    m->dm.anc = (const int*)put(anc.data(), anc.size() * sizeof(int));
    for (int c = 0; c < 3; ++c) m->dm.grav[c] = d->grav[c];
    e = hipMemcpy(m->dbuf, host.data(), bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(m->dbuf); delete m; return fail(RMX_E_HIP, std::string("hipMemcpy(model): ") + hipGetErrorString(e)); }
    *out = m;
    return RMX_OK;
}

extern "C" void rmx_model_destroy(rmx_model* m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    if (m->dbuf) (void)hipFree(m->dbuf);
    if (m->dcon) (void)hipFree(m->dcon);
    delete m;
}

// scene.forces{end+1} = ForceGroundCuboid(body); setTransform / setStiffness / setDamping / setFriction
// (scenesRedMax.m:303-309, ForceGroundCuboid.m:18-48) for every flagged body, one ground frame per scene.
extern "C" int rmx_model_set_ground_contact(rmx_model* m, const rmx_ground_contact* gc) {
    if (!m || !gc || !gc->flags || !gc->sides) return fail(RMX_E_INVALID, "null argument");
    if (!(gc->kn >= 0) || !(gc->kt >= 0) || !(gc->mu >= 0) || !(gc->kd >= 0)) return fail(RMX_E_INVALID, "contact constants must be >= 0");
    HIPCHK(hipSetDevice(m->device));
    std::vector<double> con(4 * MAXN, 0.0);
    bool any = false;
    for (int L = 0; L < m->n; ++L) {
        const int k = m->node_of_listing[L];
        con[k] = gc->flags[L] ? 1.0 : 0.0;
        any = any || gc->flags[L];
        for (int c = 0; c < 3; ++c) con[(1 + c) * MAXN + k] = gc->sides[3 * L + c];
    }
    if (m->dcon) { (void)hipFree(m->dcon); m->dcon = nullptr; m->dm.con = nullptr; }
    if (!any) return RMX_OK;
    HIPCHK(hipMalloc(&m->dcon, con.size() * sizeof(double)));
    HIPCHK(hipMemcpy(m->dcon, con.data(), con.size() * sizeof(double), hipMemcpyHostToDevice));
    m->dm.con = (const double*)m->dcon;
    const M4 E = from_cm(gc->E);
    for (int c = 0; c < 3; ++c) {
        m->dm.gn[c] = E.a[c][2];      // ng = E(1:3,3)   ForceGroundCuboid.m:70
        m->dm.gx[c] = E.a[c][3];      // xg = E(1:3,4)   :69
    }
    m->dm.kn = gc->kn; m->dm.kt = gc->kt; m->dm.mu = gc->mu; m->dm.kdc = gc->kd;
    return RMX_OK;
}
extern "C" int rmx_model_nr(const rmx_model* m) { return m ? m->nr : RMX_E_INVALID; }
extern "C" int rmx_model_nm(const rmx_model* m) { return m ? m->nm : RMX_E_INVALID; }
extern "C" int rmx_model_idxR(const rmx_model* m, int* idx) {
    if (!m || !idx) return fail(RMX_E_INVALID, "null argument");
    for (int i = 0; i < m->n; ++i) idx[i] = m->idx_listing[i];
    return RMX_OK;
}

extern "C" int rmx_batch_create(rmx_model* m, int batch, rmx_batch** out) {
    if (!m || !out || batch < 1) return fail(RMX_E_INVALID, "bad argument");
    *out = nullptr;
    HIPCHK(hipSetDevice(m->device));
    rmx_batch* b = new rmx_batch();
    b->m = m;
    b->B = batch;
    const size_t nb = sizeof(double) * (size_t)batch * std::max(m->nr, 1);
    hipError_t e = hipSuccess;
    auto alloc = [&](void** p, size_t bytes) { if (e == hipSuccess) { e = hipMalloc(p, bytes); if (e == hipSuccess) e = hipMemset(*p, 0, bytes); } };
    alloc((void**)&b->q, nb); alloc((void**)&b->qd, nb); alloc((void**)&b->qp, nb); alloc((void**)&b->qdp, nb);
    alloc((void**)&b->tmpA, nb); alloc((void**)&b->tmpB, nb); alloc((void**)&b->tmpC, nb);
    alloc((void**)&b->started, sizeof(int));
    alloc((void**)&b->it, sizeof(int) * batch); alloc((void**)&b->ls, sizeof(int) * batch); alloc((void**)&b->status, sizeof(int) * batch);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&b->ev0);
    if (e == hipSuccess) e = hipEventCreate(&b->ev1);
    if (e != hipSuccess) {
        std::string msg = std::string("rmx_batch_create: ") + hipGetErrorString(e);
        rmx_batch_destroy(b);
        return fail(RMX_E_HIP, msg);
    }
    *out = b;
    return RMX_OK;
}

extern "C" void rmx_batch_destroy(rmx_batch* b) {
    if (!b) return;
    (void)hipSetDevice(b->m->device);
    if (b->stream) (void)hipStreamSynchronize(b->stream);
    for (void* p : {(void*)b->q, (void*)b->qd, (void*)b->qp, (void*)b->qdp, (void*)b->tmpA, (void*)b->tmpB, (void*)b->tmpC,
                    (void*)b->started, (void*)b->it, (void*)b->ls, (void*)b->status})
        if (p) (void)hipFree(p);
    if (b->ev0) (void)hipEventDestroy(b->ev0);
    if (b->ev1) (void)hipEventDestroy(b->ev1);
    if (b->stream) (void)hipStreamDestroy(b->stream);
    delete b;
}
extern "C" int rmx_batch_size(const rmx_batch* b) { return b ? b->B : RMX_E_INVALID; }
extern "C" void* rmx_batch_stream(const rmx_batch* b) { return b ? (void*)b->stream : nullptr; }
extern "C" double rmx_last_step_ms(const rmx_batch* b) { return b ? b->last_ms : -1.0; }

static int copy_state(rmx_batch* b, const double* q, const double* qd, hipMemcpyKind kind, bool set) {
    if (!b) return fail(RMX_E_INVALID, "null batch");
    HIPCHK(hipSetDevice(b->m->device));
    const size_t nb = sizeof(double) * (size_t)b->B * b->m->nr;
    if (nb == 0) return RMX_OK;
    if (set) {
        if (q) HIPCHK(hipMemcpyAsync(b->q, q, nb, kind, b->stream));
        if (qd) HIPCHK(hipMemcpyAsync(b->qd, qd, nb, kind, b->stream));
        HIPCHK(hipMemsetAsync(b->started, 0, sizeof(int), b->stream));   // a new state restarts BDF2 with SDIRK2
    } else {
        if (q) HIPCHK(hipMemcpyAsync((void*)q, b->q, nb, kind, b->stream));
        if (qd) HIPCHK(hipMemcpyAsync((void*)qd, b->qd, nb, kind, b->stream));
    }
    HIPCHK(hipStreamSynchronize(b->stream));
    return RMX_OK;
}
extern "C" int rmx_set_state(rmx_batch* b, const double* q, const double* qdot) { return copy_state(b, q, qdot, hipMemcpyHostToDevice, true); }
extern "C" int rmx_get_state(rmx_batch* b, double* q, double* qdot) { return copy_state(b, q, qdot, hipMemcpyDeviceToHost, false); }
extern "C" int rmx_set_state_device(rmx_batch* b, const double* q, const double* qdot) { return copy_state(b, q, qdot, hipMemcpyDeviceToDevice, true); }
extern "C" int rmx_get_state_device(rmx_batch* b, double* q, double* qdot) { return copy_state(b, q, qdot, hipMemcpyDeviceToDevice, false); }

#define DISPATCH_NP(NPV, FN, ...)             \
    switch (NPV) {                             \
        case 4: FN<4>(__VA_ARGS__); break;     \
        case 8: FN<8>(__VA_ARGS__); break;     \
        case 16: FN<16>(__VA_ARGS__); break;   \
        case 32: FN<32>(__VA_ARGS__); break;   \
        default: FN<64>(__VA_ARGS__); break;   \
    }

template <int NP>
static void launch_eval(const rmx_model* m, const rmx_batch* b, bool wantH, double eta, double* dg, double* dH) {
    const dim3 grid(b->B), block(64);
    const bool ct = m->dm.con != nullptr;   // scenes with ForceGroundCuboid run the contact instantiations
    if (wantH && ct) k_eval<NP, true, true><<<grid, block, m->smem_bytes, b->stream>>>(m->dm, b->B, b->tmpA, b->tmpB, b->tmpC, eta, dg, dH);
    else if (wantH) k_eval<NP, true, false><<<grid, block, m->smem_bytes, b->stream>>>(m->dm, b->B, b->tmpA, b->tmpB, b->tmpC, eta, dg, dH);
    else if (ct) k_eval<NP, false, true><<<grid, block, m->smem_bytes, b->stream>>>(m->dm, b->B, b->tmpA, b->tmpB, b->tmpC, eta, dg, dH);
    else k_eval<NP, false, false><<<grid, block, m->smem_bytes, b->stream>>>(m->dm, b->B, b->tmpA, b->tmpB, b->tmpC, eta, dg, dH);
}
template <int NP>
static void launch_step_np(const rmx_model* m, const rmx_batch* b, int integ, const DevOpts& o, const StepArgs& a) {
    const dim3 grid(b->B), block(64);
    const bool ct = m->dm.con != nullptr;
    if (integ == INTEG_BDF1 && ct) k_step_bdf1<NP, true><<<grid, block, m->smem_bytes, b->stream>>>(m->dm, o, a);
    else if (integ == INTEG_BDF1) k_step_bdf1<NP, false><<<grid, block, m->smem_bytes, b->stream>>>(m->dm, o, a);
    else if (ct) k_step_bdf2<NP, true><<<grid, block, m->smem_bytes, b->stream>>>(m->dm, o, a);
    else k_step_bdf2<NP, false><<<grid, block, m->smem_bytes, b->stream>>>(m->dm, o, a);
}
template <int NP>
static void launch_euler(const rmx_model* m, const rmx_batch* b, double h, const StepArgs& a) {
    const dim3 grid(b->B), block(64);
    k_step_euler<NP><<<grid, block, m->smem_bytes, b->stream>>>(m->dm, h, a);
}
template <int NP>
static void launch_energy(const rmx_model* m, const rmx_batch* b, double* dT, double* dV) {
    const dim3 grid(b->B), block(64);
    if (m->dm.con) k_energy<NP, true><<<grid, block, m->smem_bytes, b->stream>>>(m->dm, b->B, b->q, b->qd, dT, dV);
    else k_energy<NP, false><<<grid, block, m->smem_bytes, b->stream>>>(m->dm, b->B, b->q, b->qd, dT, dV);
}

extern "C" int rmx_eval(rmx_batch* b, const double* q, const double* qA, const double* qB, double eta, double* g, double* H) {
    if (!b || !q || !qA || !qB || !g) return fail(RMX_E_INVALID, "null argument");
    if (!(eta > 0)) return fail(RMX_E_INVALID, "eta must be positive");
    rmx_model* m = b->m;
    HIPCHK(hipSetDevice(m->device));
    const size_t nv = (size_t)b->B * m->nr;
    if (nv == 0) return RMX_OK;
    double *dg = nullptr, *dH = nullptr;
    HIPCHK(hipMemcpyAsync(b->tmpA, q, nv * sizeof(double), hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipMemcpyAsync(b->tmpB, qA, nv * sizeof(double), hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipMemcpyAsync(b->tmpC, qB, nv * sizeof(double), hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipMalloc((void**)&dg, nv * sizeof(double)));
    if (H) {
        hipError_t e = hipMalloc((void**)&dH, nv * m->nr * sizeof(double));
        if (e != hipSuccess) { (void)hipFree(dg); return fail(RMX_E_NOMEM, "hipMalloc(H)"); }
        (void)hipMemsetAsync(dH, 0, nv * m->nr * sizeof(double), b->stream);
    }
    DISPATCH_NP(m->NP, launch_eval, m, b, H != nullptr, eta, dg, dH);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(g, dg, nv * sizeof(double), hipMemcpyDeviceToHost, b->stream);
    if (e == hipSuccess && H) e = hipMemcpyAsync(H, dH, nv * m->nr * sizeof(double), hipMemcpyDeviceToHost, b->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
    (void)hipFree(dg);
    if (dH) (void)hipFree(dH);
    if (e != hipSuccess) return fail(RMX_E_HIP, std::string("rmx_eval: ") + hipGetErrorString(e));
    return RMX_OK;
}

static int make_opts(const rmx_batch* b, const rmx_opts* o, DevOpts& d) {
    rmx_opts def;
    rmx_opts_default(&def);
    if (!o) o = &def;
    if (!(o->h > 0)) return fail(RMX_E_INVALID, "opts.h must be positive");
    d.h = o->h;
    d.tol = o->tol;
    d.dxMax = o->dxMax;
    d.iterMax = o->iterMaxPerDof * b->m->nr;    // iterMax = 10*length(xInit) (:97)
    d.iterLsMax = o->iterLsMax;
    d.lu_mode = o->lu_mode;
    return RMX_OK;
}

static int launch_step(rmx_batch* b, const rmx_opts* opts, int nsteps, int integ, bool with_stats, double* dT, double* dV) {
    rmx_model* m = b->m;
    DevOpts o;
    int rc = make_opts(b, opts, o);
    if (rc) return rc;
    StepArgs a{};
    a.B = b->B;
    a.nsteps = nsteps;
    a.q = b->q; a.qd = b->qd; a.qp = b->qp; a.qdp = b->qdp; a.started = b->started;
    a.it = with_stats ? b->it : nullptr; a.ls = b->ls; a.status = b->status;
    a.histT = dT; a.histV = dV;
    HIPCHK(hipEventRecord(b->ev0, b->stream));
    DISPATCH_NP(m->NP, launch_step_np, m, b, integ, o, a);
    if (integ == INTEG_BDF2) HIPCHK(hipMemsetAsync(b->started, 1, sizeof(int), b->stream));   // any non-zero value
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(b->ev1, b->stream));
    return RMX_OK;
}

static int step_sync(rmx_batch* b, const rmx_opts* opts, int nsteps, rmx_stats* st, double* hT, double* hV, int integ) {
    if (!b) return fail(RMX_E_INVALID, "null batch");
    if (nsteps < 0) return fail(RMX_E_INVALID, "nsteps < 0");
    if ((hT == nullptr) != (hV == nullptr)) return fail(RMX_E_INVALID, "hist_T and hist_V must be given together");
    rmx_model* m = b->m;
    HIPCHK(hipSetDevice(m->device));
    if (nsteps == 0 || m->nr == 0) return RMX_OK;
    double *dT = nullptr, *dV = nullptr;
    const size_t nh = (size_t)nsteps * b->B;
    if (hT) {
        HIPCHK(hipMalloc((void**)&dT, nh * sizeof(double)));
        hipError_t e = hipMalloc((void**)&dV, nh * sizeof(double));
        if (e != hipSuccess) { (void)hipFree(dT); return fail(RMX_E_NOMEM, "hipMalloc(hist)"); }
    }
    const bool ws = st != nullptr;
    if (ws) {
        (void)hipMemsetAsync(b->it, 0, sizeof(int) * b->B, b->stream);
        (void)hipMemsetAsync(b->ls, 0, sizeof(int) * b->B, b->stream);
        (void)hipMemsetAsync(b->status, 0, sizeof(int) * b->B, b->stream);
    }
    int rc = launch_step(b, opts, nsteps, integ, ws, dT, dV);
    hipError_t e = hipSuccess;
    if (rc == RMX_OK) {
        if (hT) {
            e = hipMemcpyAsync(hT, dT, nh * sizeof(double), hipMemcpyDeviceToHost, b->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(hV, dV, nh * sizeof(double), hipMemcpyDeviceToHost, b->stream);
        }
        if (e == hipSuccess && ws) {
            if (st->newton_iters) e = hipMemcpyAsync(st->newton_iters, b->it, sizeof(int) * b->B, hipMemcpyDeviceToHost, b->stream);
            if (e == hipSuccess && st->ls_halvings) e = hipMemcpyAsync(st->ls_halvings, b->ls, sizeof(int) * b->B, hipMemcpyDeviceToHost, b->stream);
            if (e == hipSuccess && st->status) e = hipMemcpyAsync(st->status, b->status, sizeof(int) * b->B, hipMemcpyDeviceToHost, b->stream);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
        if (e == hipSuccess) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, b->ev0, b->ev1) == hipSuccess) b->last_ms = ms;
        }
    }
    if (dT) (void)hipFree(dT);
    if (dV) (void)hipFree(dV);
    if (rc) return rc;
    if (e != hipSuccess) return fail(RMX_E_HIP, std::string("rmx_step: ") + hipGetErrorString(e));
    return RMX_OK;
}

extern "C" int rmx_step_bdf1(rmx_batch* b, const rmx_opts* opts, int nsteps, rmx_stats* stats, double* hist_T, double* hist_V) {
    return step_sync(b, opts, nsteps, stats, hist_T, hist_V, INTEG_BDF1);
}
extern "C" int rmx_step_bdf2(rmx_batch* b, const rmx_opts* opts, int nsteps, rmx_stats* stats, double* hist_T, double* hist_V) {
    return step_sync(b, opts, nsteps, stats, hist_T, hist_V, INTEG_BDF2);
}

extern "C" int rmx_step_euler(rmx_batch* b, double h, int nsteps, double* hT, double* hV) {
    if (!b) return fail(RMX_E_INVALID, "null batch");
    if (b->m->dm.con) return fail(RMX_E_INVALID, "rmx_step_euler: matlab-simple has no ForceGroundCuboid; use rmx_step_bdf1/bdf2");
    if (nsteps < 0 || !(h > 0)) return fail(RMX_E_INVALID, "bad nsteps / h");
    if ((hT == nullptr) != (hV == nullptr)) return fail(RMX_E_INVALID, "hist_T and hist_V must be given together");
    rmx_model* m = b->m;
    HIPCHK(hipSetDevice(m->device));
    if (nsteps == 0 || m->nr == 0) return RMX_OK;
    double *dT = nullptr, *dV = nullptr;
    const size_t nh = (size_t)nsteps * b->B;
    if (hT) {
        HIPCHK(hipMalloc((void**)&dT, nh * sizeof(double)));
        hipError_t e = hipMalloc((void**)&dV, nh * sizeof(double));
        if (e != hipSuccess) { (void)hipFree(dT); return fail(RMX_E_NOMEM, "hipMalloc(hist)"); }
    }
    StepArgs a{};
    a.B = b->B; a.nsteps = nsteps; a.q = b->q; a.qd = b->qd; a.histT = dT; a.histV = dV;
    hipError_t e = hipEventRecord(b->ev0, b->stream);
    DISPATCH_NP(m->NP, launch_euler, m, b, h, a);
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipEventRecord(b->ev1, b->stream);
    if (e == hipSuccess) e = hipMemsetAsync(b->started, 0, sizeof(int), b->stream);
    if (e == hipSuccess && hT) {
        e = hipMemcpyAsync(hT, dT, nh * sizeof(double), hipMemcpyDeviceToHost, b->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(hV, dV, nh * sizeof(double), hipMemcpyDeviceToHost, b->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
    if (e == hipSuccess) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, b->ev0, b->ev1) == hipSuccess) b->last_ms = ms;
    }
    if (dT) (void)hipFree(dT);
    if (dV) (void)hipFree(dV);
    if (e != hipSuccess) return fail(RMX_E_HIP, std::string("rmx_step_euler: ") + hipGetErrorString(e));
    return RMX_OK;
}

template <int NP>
static void launch_adjoint(const rmx_model* m, const rmx_batch* b, const DevOpts& o, const AdjArgs& a) {
    const dim3 grid(b->B), block(64);
    k_adjoint_fwd<NP><<<grid, block, m->smem_bytes, b->stream>>>(m->dm, o, a);
    k_adjoint_bwd<NP><<<grid, block, 0, b->stream>>>(m->dm, o, a);
}

extern "C" int rmx_adjoint_bdf1(rmx_batch* b, const rmx_opts* opts, int nsteps, const rmx_task_pointpos* task, const double* p,
                                double* P, double* dPdp, rmx_stats* stats) {
    if (!b || !task || !p || !P || !dPdp) return fail(RMX_E_INVALID, "null argument");
    rmx_model* m = b->m;
    if (nsteps < 1) return fail(RMX_E_INVALID, "nsteps < 1");
    if (m->dm.con) return fail(RMX_E_INVALID, "rmx_adjoint_bdf1: ground contact is outside the adjoint path (SURVEY.md 8(f))");
    if (task->body < 0 || task->body >= m->n) return fail(RMX_E_INVALID, "task body out of range");
    if (task->step < 1 || task->step > nsteps) return fail(RMX_E_INVALID, "task step must be in [1, nsteps]");
    HIPCHK(hipSetDevice(m->device));
    DevOpts o;
    int rc = make_opts(b, opts, o);
    if (rc) return rc;
    const size_t nn = (size_t)m->n * m->n, nv = (size_t)b->B * m->nr;
    const size_t hist = (size_t)b->B * nsteps * nn * sizeof(double);
    if (3 * hist > ((size_t)200 << 30)) return fail(RMX_E_NOMEM, "adjoint history (H, M, D per step) would exceed 200 GiB");
    AdjArgs a{};
    a.B = b->B; a.nsteps = nsteps; a.task_step = task->step; a.task_node = m->node_of_listing[task->body];
    for (int c = 0; c < 3; ++c) { a.xl[c] = task->xlocal[c]; a.xt[c] = task->xtarget[c]; }
    a.pscale = task->pscale; a.wreg = task->wreg; a.wpos = task->wpos;
    a.q = b->q; a.qd = b->qd; a.p = b->tmpA;
    a.it = stats ? b->it : nullptr; a.status = b->status;
    void* bufs[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    const size_t sizes[6] = {hist, hist, hist, (size_t)b->B * m->n * sizeof(double), (size_t)b->B * sizeof(double), nv * sizeof(double)};
    hipError_t e = hipSuccess;
    for (int i = 0; i < 6 && e == hipSuccess; ++i) e = hipMalloc(&bufs[i], sizes[i] ? sizes[i] : 8);
    if (e == hipSuccess) e = hipMemsetAsync(bufs[3], 0, sizes[3], b->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(b->tmpA, p, nv * sizeof(double), hipMemcpyHostToDevice, b->stream);
    if (e == hipSuccess) {
        a.Hs = (double*)bufs[0]; a.Ms = (double*)bufs[1]; a.Ds = (double*)bufs[2];
        a.dPdq = (double*)bufs[3]; a.P = (double*)bufs[4]; a.dPdp = (double*)bufs[5];
        e = hipEventRecord(b->ev0, b->stream);
        DISPATCH_NP(m->NP, launch_adjoint, m, b, o, a);
        if (e == hipSuccess) e = hipGetLastError();
        if (e == hipSuccess) e = hipEventRecord(b->ev1, b->stream);
        if (e == hipSuccess) e = hipMemsetAsync(b->started, 0, sizeof(int), b->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(P, a.P, sizes[4], hipMemcpyDeviceToHost, b->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(dPdp, a.dPdp, sizes[5], hipMemcpyDeviceToHost, b->stream);
        if (e == hipSuccess && stats) {
            if (stats->newton_iters) e = hipMemcpyAsync(stats->newton_iters, b->it, sizeof(int) * b->B, hipMemcpyDeviceToHost, b->stream);
            if (e == hipSuccess && stats->status) e = hipMemcpyAsync(stats->status, b->status, sizeof(int) * b->B, hipMemcpyDeviceToHost, b->stream);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
        if (e == hipSuccess) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, b->ev0, b->ev1) == hipSuccess) b->last_ms = ms;
        }
    }
    for (void* ptr : bufs)
        if (ptr) (void)hipFree(ptr);
    if (e != hipSuccess) return fail(RMX_E_HIP, std::string("rmx_adjoint_bdf1: ") + hipGetErrorString(e));
    return RMX_OK;
}

extern "C" int rmx_step_bdf1_async(rmx_batch* b, const rmx_opts* opts, int nsteps) {
    if (!b) return fail(RMX_E_INVALID, "null batch");
    if (nsteps <= 0 || b->m->nr == 0) return RMX_OK;
    HIPCHK(hipSetDevice(b->m->device));
    return launch_step(b, opts, nsteps, INTEG_BDF1, true, nullptr, nullptr);   // counters accumulate on the device
}
extern "C" int rmx_stats_reset(rmx_batch* b) {
    if (!b) return fail(RMX_E_INVALID, "null batch");
    HIPCHK(hipSetDevice(b->m->device));
    HIPCHK(hipMemsetAsync(b->it, 0, sizeof(int) * b->B, b->stream));
    HIPCHK(hipMemsetAsync(b->ls, 0, sizeof(int) * b->B, b->stream));
    HIPCHK(hipMemsetAsync(b->status, 0, sizeof(int) * b->B, b->stream));
    return RMX_OK;
}
extern "C" int rmx_stats_read(rmx_batch* b, rmx_stats* st) {
    if (!b || !st) return fail(RMX_E_INVALID, "null argument");
    HIPCHK(hipSetDevice(b->m->device));
    if (st->newton_iters) HIPCHK(hipMemcpyAsync(st->newton_iters, b->it, sizeof(int) * b->B, hipMemcpyDeviceToHost, b->stream));
    if (st->ls_halvings) HIPCHK(hipMemcpyAsync(st->ls_halvings, b->ls, sizeof(int) * b->B, hipMemcpyDeviceToHost, b->stream));
    if (st->status) HIPCHK(hipMemcpyAsync(st->status, b->status, sizeof(int) * b->B, hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    return RMX_OK;
}
extern "C" int rmx_sync(rmx_batch* b) {
    if (!b) return fail(RMX_E_INVALID, "null batch");
    HIPCHK(hipSetDevice(b->m->device));
    HIPCHK(hipStreamSynchronize(b->stream));
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, b->ev0, b->ev1) == hipSuccess) b->last_ms = ms;
    return RMX_OK;
}

template <int NP>
static void launch_phase(const rmx_model* m, const rmx_batch* b, int reps, double h, unsigned long long* d) {
    const dim3 grid(b->B), block(64);
    k_phase_time<NP><<<grid, block, m->smem_bytes, b->stream>>>(m->dm, reps, b->q, b->qd, h, d);
}
extern "C" int rmx_profile_phases(rmx_batch* b, int reps, double h, double* cycles4) {
    if (!b || !cycles4 || reps < 1) return fail(RMX_E_INVALID, "bad argument");
    rmx_model* m = b->m;
    HIPCHK(hipSetDevice(m->device));
    unsigned long long* d = nullptr;
    HIPCHK(hipMalloc((void**)&d, sizeof(unsigned long long) * 16 * b->B));
    DISPATCH_NP(m->NP, launch_phase, m, b, reps, h, d);
    std::vector<unsigned long long> hbuf(16 * (size_t)b->B);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(hbuf.data(), d, sizeof(unsigned long long) * hbuf.size(), hipMemcpyDeviceToHost, b->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(RMX_E_HIP, std::string("rmx_profile_phases: ") + hipGetErrorString(e));
    for (int k = 0; k < 16; ++k) {   // [0..3] phase totals, [4..15] stamps inside the (g,H) evaluation
        double sum = 0;
        for (int t = 0; t < b->B; ++t) sum += (double)hbuf[16 * (size_t)t + k];
        cycles4[k] = sum / ((double)b->B * reps);
    }
    return RMX_OK;
}

extern "C" int rmx_energy(rmx_batch* b, double* T, double* V) {
    if (!b || !T || !V) return fail(RMX_E_INVALID, "null argument");
    rmx_model* m = b->m;
    HIPCHK(hipSetDevice(m->device));
    double *dT = nullptr, *dV = nullptr;
    HIPCHK(hipMalloc((void**)&dT, sizeof(double) * b->B));
    hipError_t e = hipMalloc((void**)&dV, sizeof(double) * b->B);
    if (e != hipSuccess) { (void)hipFree(dT); return fail(RMX_E_NOMEM, "hipMalloc(energy)"); }
    DISPATCH_NP(m->NP, launch_energy, m, b, dT, dV);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(T, dT, sizeof(double) * b->B, hipMemcpyDeviceToHost, b->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(V, dV, sizeof(double) * b->B, hipMemcpyDeviceToHost, b->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
    (void)hipFree(dT);
    (void)hipFree(dV);
    if (e != hipSuccess) return fail(RMX_E_HIP, std::string("rmx_energy: ") + hipGetErrorString(e));
    return RMX_OK;
}
