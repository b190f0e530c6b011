// rmx_kernels.hip -- kernel entry points for gfx950 and their launchers, for ONE padded tree size RMX_NP (lanes in use per
// wavefront: 4, 8, 16, 32 or 64).  __graft_entry__.build() compiles this file once per size, in parallel, and links the
// objects with redmax_hip.hip (host side / C ABI).  Device code: rmx_device.h.
#ifndef RMX_NP
#error "compile with -DRMX_NP=4|8|16|32|64"
#endif
#include <algorithm>
#include <cstdlib>

#include "rmx_host.h"

// ============================================================================ kernels



// the per-node constants in the layout the evaluation reads them in, [NCONST][cstride(NP)] (see eval_front_e2): into the wavefront's
// LDS (smem_setup) or, once per model, into a global table (k_stage_consts for the RMX_GLOBAL_CONSTS kernels)
template <int NP>
__device__ __forceinline__ void stage_consts(const DevModel& M, double* __restrict__ dst) {
    constexpr int CS = cstride(NP);
    if (threadIdx.x < CS) {       // column NP (trees padded to < 64 lanes) and the unused node slots n..NP-1: idle defaults
        const int j = threadIdx.x;
        const bool in = j < M.n;
        double* c = dst;
        for (int r = 0; r < 36; ++r) c[r * CS + j] = in ? M.K[r * MAXN + j] : ((r == 0 || r == 4 || r == 8) ? 1.0 : 0.0);   // identity
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
        for (int r = 0; r < 4; ++r) c[r * CS + j] = (in && M.con) ? M.con[r * MAXN + j] : 0.0;
    }
}

template <int NP>
__device__ __forceinline__ void smem_setup(const DevModel& M, double*& sAcc, double*& sCol) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    sAcc = smem;
    if (threadIdx.x < ACC_STRIDE) sAcc[M.n * ACC_STRIDE + threadIdx.x] = 0.0;   // zero row n (end-of-tree suffix)
#ifdef RMX_GLOBAL_CONSTS
    sCol = const_cast<double*>(M.gconst);     // read-only here: the kernels of this translation unit never switch Euler charts
#else
    sCol = smem + acc_doubles(M.n, NP);       // per-node constants, [NCONST][NP] (see eval_front_e2)
    stage_consts<NP>(M, sCol);
#endif
    __syncthreads();
}

// The contact-capable kernels: every body's own ground frame and constants (DevModel::con rows 4..13, GroundC) behind the per-node
// constants.  Ends with a barrier.
template <int NP>
__device__ __forceinline__ void con_setup(const DevModel& M, double* __restrict__ sCol) {
    constexpr int CS = cstride(NP);
    if (M.con && threadIdx.x < CS) {
        const int j = threadIdx.x;
        const bool in = j < M.n;
        for (int r = 0; r < NGROUND; ++r) sCol[(NCONST + r) * CS + j] = in ? M.con[(4 + r) * MAXN + j] : 0.0;
    }
    __syncthreads();
}

template <int NP>
__global__ void __launch_bounds__(64) k_stage_consts(const DevModel M, double* __restrict__ dst) {
    stage_consts<NP>(M, dst);
}

// simLoop (driverRedMaxBDF1.m:57-91): all steps of one trajectory inside one wavefront.
// Contact-capable scenes (CT) take two launches per call.  LEAN: the plain evaluation plus the test that every cuboid is clear of
// the ground; a trajectory that fails it stops at the start of that step and leaves the step index in a.resume.  The second
// launch (CT, not LEAN: the evaluation with the contact terms) takes every trajectory from its a.resume to the end.  In one
// kernel the contact terms' registers (72 accumulators for the K/D blocks on top of the Hessian stage's state) put the whole
// Newton loop into scratch - 34 us per iteration in free flight against 12 us for the plain kernel (tools/ct_overhead.py) -
// and an out-of-line call of the contact solve tripled ITS cost (the model constants arrive as a pointer into scratch).
// FULLCHAIN: an instantiation for serial chains that fill every node slot (is_chain && n == NP, decided by the launcher): the two
// model facts are compile-time constants there, so the tree paths (pointer jumping, relation masks, subtree ranges) and the
// per-row bounds of partly filled sizes are not even compiled in - fewer live masks, a smaller loop body.
// TAG >= TAG_FULLN: a tree (not necessarily a chain) that fills every node slot: n == NP at compile time, so the per-row bounds of partly
// filled sizes go (the 64-joint tree of BASELINE.json configs[2]).
constexpr int TAG_FULLN = 4;
constexpr int TAG_COOP = 8;      // k_step_bdf1/2<32, true, false, false, TAG_COOP>: the cooperative launch (RMX_PART 4)
constexpr int TAG_W2 = 16;       // k_step_bdf1/2<64, false, false, false, TAG_W2>: two wavefronts per 64-node tree (RMX_PART 5, RMX_W2);
                                 // TAG_W2 + 1: the same for trees of 33..63 nodes (n at run time)
constexpr int TAG_W2_NOE = TAG_W2 + 2;      // TAG_W2 for a call that records no energies (the benchmark's launch): the energies of the last
                                             // evaluation are neither copied per Newton iteration nor summed by the helper wave
constexpr bool tag_w2(const int tag) { return tag == TAG_W2 || tag == TAG_W2 + 1 || tag == TAG_W2_NOE; }
constexpr bool tag_w2_full(const int tag) { return tag == TAG_W2 || tag == TAG_W2_NOE; }
template <int NP, bool FULLCHAIN, int TAG = 0>
__device__ __forceinline__ DevModel model_view(const DevModel& Min) {
    DevModel M = Min;
    if constexpr (FULLCHAIN) {
        M.n = NP;
        M.is_chain = 1;
    }
    if constexpr ((TAG >= TAG_FULLN && TAG < TAG_COOP) || tag_w2_full(TAG)) M.n = NP;
    return M;
}

template <int NP>
__device__ __forceinline__ double* w2_help_area(const DevModel& M, double* sAcc) {      // (W2_HELP_*: behind the per-node constants)
    return const_cast<double*>(static_cast<const double*>(RMX_CONSTS(sAcc, M.n, NP))) + (NCONST + NGROUND) * cstride(NP);
}
// RMX_W2: wave 1 of a two-wave workgroup.  It serves wave 0's guarded Newton iterations - the odd columns of the Hessian tiles, its
// share of the block-column elimination - and sleeps at the workgroup barrier in between (eval_hess and lu_solve_neg_diag64_staged
// hold wave 0's side of the same barriers).
template <int NP, bool ENERGY = true>
__device__ __forceinline__ void w2_helper(const DevModel& M, double* __restrict__ sAcc, const int lane) {
    if constexpr (NP == 64 && RMX_W2) {
        constexpr int CS = cstride(NP);
        const double* cRel = RMX_CONSTS(sAcc, M.n, NP) + (36 + 6 + 4 + 8 + 1) * CS;
        double* const hp = w2_help_area<NP>(M, sAcc);
        if (lane < W2_HELP_AS) hp[M.n * W2_HELP_AS + lane] = 0.0;                    // (row n of an accumulation scratch stays zero)
        for (;;) {
            RMX_WG_BAR();
            const int cmd = *w2_cmd();
            if (cmd == 0) break;
            if (cmd == 2) {       // an evaluation that may end wave 0's solve: residual and energies only, on this wave's scratch
                const double x = hp[W2_HELP_ARGS + lane], xqd = hp[W2_HELP_ARGS + 64 + lane], xv = hp[W2_HELP_ARGS + 128 + lane];
                const double eta = hp[W2_HELP_ARGS + 192];
                NodeOut e;
                FrontState f2;
                eval_front_e2<NP, false, false, false, false, W2_HELP_AS>(M, hp, lane, x, xqd, xv, eta, eta * eta, e, f2);
                const double gn2 = wave_sum_np<NP>(e.g * e.g);
                double T = 0.0, V = 0.0;
                if constexpr (ENERGY) {
                    T = wave_sum(e.eT);
                    V = wave_sum(e.eV);
                }
                if (lane == 0) {
                    hp[W2_HELP_RES] = gn2;
                    hp[W2_HELP_RES + 1] = T;
                    hp[W2_HELP_RES + 2] = V;
                }
                RMX_WG_BAR();
                continue;
            }
            double h1[4][2][4];
            hess64_tiles<NP, 1>(lane, sAcc, cRel, h1);
            RMX_WG_BAR();
            w2_store_half<1>(sAcc, lane, h1);
            RMX_WG_BAR();
#if RMX_W2
            if (M.tree_dmax == 0) (void)w2_lu_call();      // (a branching tree's solve runs along the tree, on wave 0 alone: tree_solve64)
#endif
        }
    }
}
// RMX_W2, chains of <= 32 nodes (RMX_PART 6): the helper wave only evaluates - the point wave 0 posts, with the FULL front (see W2C_HELP_*)
template <int NP>
__device__ __forceinline__ void w2c_helper(const DevModel& M, double* __restrict__ sAcc, const int lane) {
    if constexpr (NP == 32 && RMX_W2) {
        double* const hp = w2_help_area<NP>(M, sAcc);
        if (lane < ACC_STRIDE) hp[M.n * ACC_STRIDE + lane] = 0.0;                    // (row n of an accumulation scratch stays zero)
        for (;;) {
            RMX_WG_BAR();
            if (*w2_cmd() == 0) break;
            const double x = hp[W2C_HELP_ARGS + lane], xqd = hp[W2C_HELP_ARGS + 64 + lane], xv = hp[W2C_HELP_ARGS + 128 + lane];
            const double eta = hp[W2C_HELP_ARGS + 192];
            NodeOut e;
            FrontState f2;
            eval_front_e2<NP, true, false, false, false>(M, hp, lane, x, xqd, xv, eta, eta * eta, e, f2);
            const double gn2 = wave_sum_np<NP>(e.g * e.g), T = wave_sum(e.eT), V = wave_sum(e.eV);
            if (lane == 0) {
                hp[W2C_HELP_RES] = gn2;
                hp[W2C_HELP_RES + 1] = T;
                hp[W2C_HELP_RES + 2] = V;
            }
            RMX_WG_BAR();
        }
    }
}

#if RMX_W2
// BDF1 steps s, s + 1, ... of a tree under the guarded Newton (driverRedMaxBDF1.m:57-157), ONE loop around one call site of the front:
// newton_rot<NP, false> and the step epilogue of k_step_bdf1, decision for decision and operation for operation, plus this.  The
// evaluation that ENDS a solve (the accepted line-search trial whose |g| is below tol) is followed by the first evaluation of the NEXT
// step, at a point that depends on the converged iterate alone - known before that last evaluation is run.  When the current Newton
// iteration is the one the previous solve ended at, wave 0 hands the trial point to the helper wave (w2_helper, command 2: a
// residual-only front on a scratch of its own - the same operations, so the same residual) and evaluates the next step's first point
// itself; if the helper's |g|^2 ends the solve, the epilogue of the step runs here and the loop goes on as the next step's solve with
// its first evaluation done; if not, the trial point is evaluated after all (the prediction cost one evaluation).  Returns the number
// of steps finished (>= 1); ends at the first solve that ends on an evaluation of its own.
template <int NP, bool ENERGY = true>
__device__ __forceinline__ int w2_steps_bdf1(const DevModel& M, const DevOpts& o, const StepArgs& a, double* sAcc, const int lane,
                                             const int traj, const int id, const size_t off, int s, double& q, double& qd, int& iters,
                                             int& halvings, int& status, PivotPolicy& piv, int& predict) {
    double Hrow[NP];
    FrontState fs;
    NodeOut e, e0, last;
    double q0 = q;
    double x = q0 + o.h * qd;                    // initial guess (:70) and q0 + h qdot0 of dqtmp (:169)
    double qA = q0, qB = x;
    const double eta = o.h;
    double lo = 0.0, dx = 0.0, alpha = 1.0, f0 = 0.0, g0n2 = 0.0, x0 = x, lo0 = 0.0;
    int iter = 1, lsfail = 0, iterLs = 1, done = 0;
    bool ls = false, spec_failed = false;
    e0.g = e0.eT = e0.eV = 0.0;
    last = e0;
    double* const hp = w2_help_area<NP>(M, sAcc);
    constexpr int HARGS = NP == 64 ? W2_HELP_ARGS : W2C_HELP_ARGS, HRES = NP == 64 ? W2_HELP_RES : W2C_HELP_RES;
    // the epilogue of step s (k_step_bdf1): qdot (:72), q, Scene.saveHistory; T, V: the sums of the last evaluation's energies
    auto finish = [&](const double T, const double V) {
        qd = ((x - q0) + lo) / o.h;
        q = x;
        if (ENERGY && a.histT && lane == 0) {
            a.histT[(size_t)s * a.B + traj] = T;
            a.histV[(size_t)s * a.B + traj] = V;
        }
        if (a.histQ && id >= 0) {
            a.histQ[(size_t)s * a.B * M.nr + off] = q;
            a.histQd[(size_t)s * a.B * M.nr + off] = qd;
        }
        ++s;
        ++done;
    };
    while (true) {
#ifndef RMX_W2_SPEC
#define RMX_W2_SPEC 1      // 0: measurement aid, the loop without the run-ahead
#endif
                const bool spec = RMX_W2_SPEC && !spec_failed && ls && iter == predict && piv.streak == 0 && s + 1 < a.nsteps && !a.w2_noahead;      // (wave-uniform)
        // (the run-ahead swaps the next step's point INTO x, lo, qA, qB: the expressions handed to the front stay newton_rot's)
        const double sx = x, slo = lo, sqA = qA, sqB = qB;
        if (spec) {
            hp[HARGS + lane] = x;
            hp[HARGS + 64 + lane] = ((x - qA) + lo) / eta;
            hp[HARGS + 128 + lane] = (x - qB) + lo;
            if (lane == 0) {
                hp[HARGS + 192] = eta;
                *w2_cmd() = 2;
            }
            RMX_WG_BAR();
            const double qn = x, qdn = ((x - q0) + lo) / o.h;      // q, qdot as `finish` forms them
            x = qn + o.h * qdn;
            lo = 0.0;
            qA = qn;
            qB = x;
        }
        eval_front<NP, true, false, false, false>(M, sAcc, lane, x, ((x - qA) + lo) / eta, (x - qB) + lo, eta, e, fs);
        if (spec) {
            RMX_WG_BAR();
            const double gh = hp[HRES], Th = hp[HRES + 1], Vh = hp[HRES + 2];
            const double nx = x;
            x = sx; lo = slo; qA = sqA; qB = sqB;
            if (!((0.5 * gh < f0 || !(iterLs < o.iterLsMax)) && sqrt(gh) < o.tol)) {
                spec_failed = true;              // not the end of this solve: the trial point is evaluated here after all
                continue;
            }
            // the solve of step s ends at x (newton_rot: halvings, pivot policy; then the step's epilogue) ...
            halvings += iterLs - 1;
            predict = iter;
            pivot_policy_update(piv);
            finish(Th, Vh);
            // ... and this is the solve of the next step after its first evaluation
            q0 = q;
            x = nx;
            qA = q0;
            qB = x;
            lo = 0.0; dx = 0.0; alpha = 1.0; f0 = 0.0; g0n2 = 0.0; x0 = x; lo0 = 0.0;
            iter = 1; lsfail = 0; iterLs = 1;
            ls = false;
        }
        spec_failed = false;
        const double gn2 = wave_sum_np<NP>(e.g * e.g);
        if (ls) {                                        // this was a trial point of the line search (:124-138)
            if (!(0.5 * gn2 < f0) && iterLs < o.iterLsMax) {
                alpha *= 0.5;
                ++iterLs;
                two_sum(x0, fma(alpha, dx, lo0), x, lo);
                lo *= o.comp;
                if (__all(x == x0 && lo == lo0)) {        // see newton_impl: every further halving re-evaluates g(x0)
                    if constexpr (ENERGY) last = e0;
                    halvings += o.iterLsMax - 1;
                    if (!(sqrt(g0n2) < o.tol)) status |= 2 | 8;
                    break;
                }
                continue;
            }
            if constexpr (ENERGY) last = e;
            halvings += iterLs - 1;
            if (sqrt(gn2) < o.tol) break;
            if (iter >= o.iterMax) {
                status |= 2;
                break;
            }
            lsfail += (0.5 * gn2 < f0) ? 0 : 1;
            if (o.lsFailLimit > 0 && lsfail >= o.lsFailLimit) {
                status |= 2 | ST_LS_CUT;
                break;
            }
            ++iter;
        }
        (void)eval_hess<NP, false, false, false>(M, lane, fs, Hrow, nullptr, sAcc, e.g);
        if constexpr (ENERGY) {
            e0 = e;
            last = e;
        }
        ++iters;
        {
            bool lu_ok;
            if constexpr (NP == 64) {
                if (M.tree_dmax > 0) {   // (a branching tree: along the tree, this wave alone - w2_helper skips the solve)
                    dx = tree_solve64(M, lane, sAcc, lu_ok);
                } else {
                    const W2Lu r = w2_lu_call();
                    dx = r.dx;
                    lu_ok = r.ok != 0;
                }
                RMX_SYNC();             // sAcc goes back to the front, whose subtree scan relies on a zero row n
                if (lane < ACC_STRIDE) sAcc[M.n * ACC_STRIDE + lane] = 0.0;
                RMX_SYNC();
            } else {
                static_assert(NP == 64 || (NP == 32 && LU_SPLIT32), "the sizes newton_rot solves from the staging area");
                dx = lu_solve_neg_diag32(M.n, lane, sAcc, e.g, lu_ok);
            }
            if (lu_ok) {
                piv.streak = 0;
            } else {             // growth guard tripped: redo this solve with partial pivoting (see newton_impl)
                ++piv.streak;
                status |= 16;
                NodeOut e2;
                eval_front<NP, true, false, false, false>(M, sAcc, lane, x, ((x - qA) + lo) / eta, (x - qB) + lo, eta, e2, fs);
                eval_hess<NP, false, false>(M, lane, fs, Hrow, nullptr, sAcc);
                dx = lu_solve_neg<NP>(M.n, lane, Hrow, e.g);
            }
        }
        const double dxn2 = wave_sum_np<NP>(dx * dx);
        if (!(dxn2 == dxn2)) {
            status |= 4;
            break;
        }
        if (sqrt(dxn2) > o.dxMax) {
            status |= 1;
            break;
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
            if constexpr (ENERGY) last = e0;
            halvings += o.iterLsMax - 1;
            if (!(sqrt(g0n2) < o.tol)) status |= 2 | 8;
            break;
        }
        ls = true;
    }
    // a solve that ended on an evaluation of its own
    predict = iter;
    pivot_policy_update(piv);
    if constexpr (ENERGY) finish(wave_sum(last.eT), wave_sum(last.eV));
    else finish(0.0, 0.0);
    return done;
}
#endif

__device__ __forceinline__ void w2_release(const int lane) {
    if (lane == 0) *w2_cmd() = 0;
    RMX_WG_BAR();
}

// TAG: keeps the kernel names of a translation unit compiled with other macros (RMX_GLOBAL_CONSTS) distinct (0, 3); TAG_FULLN, TAG_FULLN + 1:
// the n == NP instantiations of the two
template <int NP, bool CT, bool LEAN = false, bool FULLCHAIN = false, int TAG = 0>
__global__ void __launch_bounds__(tag_w2(TAG) ? 128 : 64) k_step_bdf1(const DevModel Min, const DevOpts o, const StepArgs a) {
    static_assert(!LEAN || CT, "the lean launch belongs to the contact-capable kernels");
    constexpr bool COOP = TAG == TAG_COOP;           // the cooperative launch (rmx_device.h CoopCtx): COOP_G workgroups per parked rollout
    static_assert(!COOP || (CT && !LEAN), "the cooperative launch belongs to the kernels with the contact terms");
    constexpr bool PARKS = CT && !LEAN && !COOP;     // the launch with the contact terms may park a rollout for the cooperative one
    const DevModel M = model_view<NP, FULLCHAIN, TAG>(Min);
    unsigned long long tick0 = __builtin_amdgcn_s_memtime();
    int traj = blockIdx.x;
    CoopCtx cx;
    cx.ticks = o.coopTicks;
    int pk = 0, npark = 1, pstride = 1;
    if constexpr (COOP) {
        pk = blockIdx.x / COOP_G;
        cx.member = blockIdx.x % COOP_G;
        cx.words = a.xch + (size_t)pk * COOP_WORDS;
        npark = a.park[0];
        pstride = a.ngroups;
        if (pk >= npark) return;
        traj = a.park[1 + pk];
    }
    const int s0 = (CT && !LEAN && a.resume) ? a.resume[traj] : 0;
    if (!COOP && s0 >= a.nsteps) return;           // the lean launch took this trajectory all the way
    double *sAcc, *sCol;
    smem_setup<NP>(M, sAcc, sCol);
    const int lane = threadIdx.x;
    if constexpr (tag_w2(TAG)) {
        if (threadIdx.x >= 64) {
            if constexpr (NP == 64) w2_helper<NP, TAG != TAG_W2_NOE>(M, sAcc, lane - 64);
            else w2c_helper<NP>(M, sAcc, lane - 64);
            return;
        }
    }
    int* const chart0 = (CT && M.nsph) ? a.chart : nullptr;
    if constexpr (CT) con_setup<NP>(M, sCol);
    const bool writer = !COOP || cx.member == 0;     // (members 1.. of a cooperative group compute, member 0 also stores)
    for (; pk < npark; pk += pstride) {              // (one pass unless COOP: group g finishes parked rollouts g, g + ngroups, ...)
    int sfirst = s0;
    if constexpr (COOP) {
        traj = a.park[1 + pk];
        sfirst = a.resume[traj];
        tick0 = __builtin_amdgcn_s_memtime();
    }
    const int id = (lane < M.n) ? M.idx[lane] : -1;
    const size_t off = (size_t)traj * M.nr + (id >= 0 ? id : 0);
    double q = id >= 0 ? a.q[off] : 0.0;
    double qd = id >= 0 ? a.qd[off] : 0.0;
    int iters = 0, halv = 0, status = 0;
    PivotPolicy piv;
    if constexpr (COOP) {
        const int* pp = a.park + 1 + a.B + 3 * traj;
        piv.hold = pp[0]; piv.len = pp[1]; piv.streak = pp[2];
    }
    int* const chart = chart0 ? chart0 + (size_t)traj * M.nsph : nullptr;
    if constexpr (CT) {
        if (M.nsph) sph_setup<NP>(M, sCol, lane, chart);
    }
    int stop = a.nsteps;
#if RMX_W2
    int w2_predict = 0;                            // (w2_steps_bdf1: the Newton iteration the last solve ended at)
#endif
    for (int s = sfirst; s < a.nsteps; ++s) {
#if RMX_W2
        if constexpr (tag_w2(TAG) && (!FULLCHAIN || NP == 32)) {
            // guarded solves of a tree: the loop that runs ahead into the next step (w2_steps_bdf1); pivoting solves and the 64-lane serial
            // chains (whose residual-only front sums by another scan) keep newton_node.  NP == 32: the chains of RMX_PART 6, whose helper
            // runs the full front
            if ((NP == 32 || !M.is_chain) && o.lu_mode == 0 && piv.hold == 0) {
                s += w2_steps_bdf1<NP, TAG != TAG_W2_NOE>(M, o, a, sAcc, lane, traj, id, off, s, q, qd, iters, halv, status, piv, w2_predict) - 1;
                continue;
            }
        }
#endif
        const double q0 = q, qd0 = qd;
        const double xg = q0 + o.h * qd0;          // initial guess (:70) and q0 + h qdot0 of dqtmp (:169)
        NodeOut last;
        double xlo;
        const int it_in = iters, hv_in = halv, st_in = status;
        const PivotPolicy piv_in = piv;
#if RMX_W2
        // (a full tree gets here for its pivoting solves only - newton_policy's first branch: the guarded loop is w2_steps_bdf1)
        double x;
        if constexpr (tag_w2_full(TAG) && (!FULLCHAIN || NP == 32)) {
            if (piv.hold > 0) --piv.hold;
            xlo = 0.0;
            x = newton_rot<NP, true>(M, o, sAcc, sCol, lane, xg, q0, xg, o.h, last, iters, halv, status, piv, xlo);
        } else {
            x = newton_node<NP, CT, LEAN, COOP>(M, o, sAcc, sCol, lane, xg, q0, xg, o.h, last, iters, halv, status, piv, xlo, cx);
        }
#else
        const double x = newton_node<NP, CT, LEAN, COOP>(M, o, sAcc, sCol, lane, xg, q0, xg, o.h, last, iters, halv, status, piv, xlo, cx);
#endif
        if (LEAN && (status & ST_LEFT_LEAN)) {
            status &= ~ST_LEFT_LEAN;
            stop = s;
            break;
        }
        if (PARKS && (status & ST_PARK)) {         // this step goes to the cooperative launch, from its start: nothing of it is kept
            iters = it_in; halv = hv_in; status = st_in; piv = piv_in;
            stop = s;
            break;
        }
        if (COOP && (status & ST_COOP_FAULT)) break;
        qd = ((x - q0) + xlo) / o.h;               // (:72), with the low-order part of the iterate the residual was evaluated at
        q = x;
        if constexpr (CT) {                        // jroot.reparam() (:78)
            double np0 = 0.0, np1 = 0.0;
            if (M.nsph && sph_reparam<NP, false>(M, sCol, lane, chart, q, qd, np0, np1)) status |= 32;
        }
        if (a.histT && writer) {                   // Scene.saveHistory (Scene.m:134-161)
            const double T = wave_sum(last.eT), V = wave_sum(last.eV);
            if (lane == 0) {
                a.histT[(size_t)s * a.B + traj] = T;
                a.histV[(size_t)s * a.B + traj] = V;
            }
        }
        if (a.histQ && id >= 0 && writer) {
            a.histQ[(size_t)s * a.B * M.nr + off] = q;
            a.histQd[(size_t)s * a.B * M.nr + off] = qd;
        }
        if constexpr (CT) {
            if (a.histC && lane < M.nsph && writer) a.histC[((size_t)s * a.B + traj) * M.nsph + lane] = chart[lane];
        }
    }
    if (id >= 0 && writer) {
        a.q[off] = q;
        a.qd[off] = qd;
    }
    if (LEAN && lane == 0) a.resume[traj] = stop;
    if constexpr (PARKS) {
        if (a.park && lane == 0) {
            a.resume[traj] = stop;
            if (stop < a.nsteps) {
                a.park[1 + atomicAdd(a.park, 1)] = traj;
                int* pp = a.park + 1 + a.B + 3 * traj;
                pp[0] = piv.hold; pp[1] = piv.len; pp[2] = piv.streak;
            }
        }
    }
    if (lane == 0 && a.it && writer) {
        a.it[traj] += iters;
        a.ls[traj] += halv;
        a.status[traj] |= status;
    }
    if (lane == 0 && a.ticks && writer) a.ticks[traj] += __builtin_amdgcn_s_memtime() - tick0;      // this rollout's share of the launch (rmx_step_ticks)
    }
    if constexpr (tag_w2(TAG)) w2_release(lane);
}

// simLoop (driverRedMaxBDF2.m:57-125): SDIRK2 start step (two Newton solves), then BDF2.  CT / LEAN: see k_step_bdf1.
template <int NP, bool CT, bool LEAN = false, bool FULLCHAIN = false, int TAG = 0>
__global__ void __launch_bounds__(tag_w2(TAG) ? 128 : 64) k_step_bdf2(const DevModel Min, const DevOpts o, const StepArgs a) {
    static_assert(!LEAN || CT, "the lean launch belongs to the contact-capable kernels");
    constexpr bool COOP = TAG == TAG_COOP;           // see k_step_bdf1
    static_assert(!COOP || (CT && !LEAN), "the cooperative launch belongs to the kernels with the contact terms");
    constexpr bool PARKS = CT && !LEAN && !COOP;
    const DevModel M = model_view<NP, FULLCHAIN, TAG>(Min);
    unsigned long long tick0 = __builtin_amdgcn_s_memtime();
    int traj = blockIdx.x;
    CoopCtx cx;
    cx.ticks = o.coopTicks;
    int pk = 0, npark = 1, pstride = 1;
    if constexpr (COOP) {
        pk = blockIdx.x / COOP_G;
        cx.member = blockIdx.x % COOP_G;
        cx.words = a.xch + (size_t)pk * COOP_WORDS;
        npark = a.park[0];
        pstride = a.ngroups;
        if (pk >= npark) return;
        traj = a.park[1 + pk];
    }
    const int s0 = (CT && !LEAN && a.resume) ? a.resume[traj] : 0;
    if (!COOP && s0 >= a.nsteps) return;
    double *sAcc, *sCol;
    smem_setup<NP>(M, sAcc, sCol);
    const int lane = threadIdx.x;
    if constexpr (tag_w2(TAG)) {
        if (threadIdx.x >= 64) {
            if constexpr (NP == 64) w2_helper<NP>(M, sAcc, lane - 64);
            else w2c_helper<NP>(M, sAcc, lane - 64);
            return;
        }
    }
    int* const chart0 = (CT && M.nsph) ? a.chart : nullptr;
    if constexpr (CT) con_setup<NP>(M, sCol);
    const bool writer = !COOP || cx.member == 0;
    const double h = o.h;
    for (; pk < npark; pk += pstride) {              // (one pass unless COOP)
    int sfirst = s0;
    if constexpr (COOP) {
        traj = a.park[1 + pk];
        sfirst = a.resume[traj];
        tick0 = __builtin_amdgcn_s_memtime();
    }
    const int id = (lane < M.n) ? M.idx[lane] : -1;
    const size_t off = (size_t)traj * M.nr + (id >= 0 ? id : 0);
    double q = id >= 0 ? a.q[off] : 0.0;
    double qd = id >= 0 ? a.qd[off] : 0.0;
    double qp = id >= 0 ? a.qp[off] : 0.0;       // step k-1 (Joint.q1 / qdot1 in the reference)
    double qdp = id >= 0 ? a.qdp[off] : 0.0;
    const bool started = (*a.started) != 0 || sfirst > 0;    // resumed behind the lean launch: its steps are this call's history
    int iters = 0, halv = 0, status = 0;
    PivotPolicy piv;
    if constexpr (COOP) {
        const int* pp = a.park + 1 + a.B + 3 * traj;
        piv.hold = pp[0]; piv.len = pp[1]; piv.streak = pp[2];
    }
    int* const chart = chart0 ? chart0 + (size_t)traj * M.nsph : nullptr;
    if constexpr (CT) {
        if (M.nsph) sph_setup<NP>(M, sCol, lane, chart);
    }
    int stop = a.nsteps;
    for (int s = sfirst; s < a.nsteps; ++s) {
        NodeOut last;
        double xlo;       // low-order part of the converged iterate: below the rounding of the multistep velocity formulas, not used
        const int it_in = iters, hv_in = halv, st_in = status;
        const PivotPolicy piv_in = piv;
        bool left = false;                         // the step was given up to the next launch (lean -> contact terms -> cooperative)
        if (s == 0 && !started) {
            const double al = (2.0 - sqrt(2.0)) / 2.0;    // (:74)
            const double q0 = q, qd0 = qd;
            // SDIRK2a (evalSDIRK2a :194-225): eta = a h, qA = q0, qB = q0 + a h qdot0
            const double xa0 = q0 + al * h * qd0;
            const double qa = newton_node<NP, CT, LEAN, COOP>(M, o, sAcc, sCol, lane, xa0, q0, q0 + (al * h) * qd0, al * h, last, iters, halv, status, piv, xlo, cx);
            left = (LEAN && (status & ST_LEFT_LEAN)) || (PARKS && (status & ST_PARK)) || (COOP && (status & ST_COOP_FAULT));
            if (!left) {
                const double qda = (qa - q0) / (al * h);
                // SDIRK2b (evalSDIRK2b :228-260)
                const double x10 = qa + (1.0 - al) * h * qda;
                const double qA = q0 + (1.0 - al) * h * qda;
                const double qB = q0 + (2.0 * al - 1.0) * h * qd0 + 2.0 * (1.0 - al) * h * qda;
                const double q1 = newton_node<NP, CT, LEAN, COOP>(M, o, sAcc, sCol, lane, x10, qA, qB, al * h, last, iters, halv, status, piv, xlo, cx);
                left = (LEAN && (status & ST_LEFT_LEAN)) || (PARKS && (status & ST_PARK)) || (COOP && (status & ST_COOP_FAULT));
                if (!left) {
                    qd = (q1 - q0 - (1.0 - al) * h * qda) / (al * h);
                    q = q1;
                    qp = q0;
                    qdp = qd0;
                }
            }
        } else {
            // BDF2 (evalBDF2 :263-293): eta = 2h/3
            const double q0 = qp, qd0 = qdp, q1 = q, qd1 = qd;
            const double x0 = q1 + h * qd1;
            const double qA = (4.0 / 3.0) * q1 - (1.0 / 3.0) * q0;
            const double qB = (4.0 / 3.0) * q1 - (1.0 / 3.0) * q0 + (8.0 / 9.0) * h * qd1 - (2.0 / 9.0) * h * qd0;
            const double q2 = newton_node<NP, CT, LEAN, COOP>(M, o, sAcc, sCol, lane, x0, qA, qB, (2.0 / 3.0) * h, last, iters, halv, status, piv, xlo, cx);
            left = (LEAN && (status & ST_LEFT_LEAN)) || (PARKS && (status & ST_PARK)) || (COOP && (status & ST_COOP_FAULT));
            if (!left) {
                qp = q1;
                qdp = qd1;
                qd = (3.0 / (2.0 * h)) * (q2 - (4.0 / 3.0) * q1 + (1.0 / 3.0) * q0);
                q = q2;
                // the Newton residual was evaluated with qdot = (q2-qA)/eta, identical up to rounding
            }
        }
        if (left) {
            if (COOP) break;                       // (ST_COOP_FAULT stays in the status)
            // nothing of this step is kept: the next launch takes it from its start (a solve of the start step that went through included)
            iters = it_in; halv = hv_in; status = st_in; piv = piv_in;
            stop = s;
            break;
        }
        if constexpr (CT) {                        // jroot.reparam() (:112): q, qdot and the previous step's q1, qdot1
            if (M.nsph && sph_reparam<NP, true>(M, sCol, lane, chart, q, qd, qp, qdp)) status |= 32;
        }
        if (a.histT && writer) {
            const double T = wave_sum(last.eT), V = wave_sum(last.eV);
            if (lane == 0) {
                a.histT[(size_t)s * a.B + traj] = T;
                a.histV[(size_t)s * a.B + traj] = V;
            }
        }
        if (a.histQ && id >= 0 && writer) {
            a.histQ[(size_t)s * a.B * M.nr + off] = q;
            a.histQd[(size_t)s * a.B * M.nr + off] = qd;
        }
        if constexpr (CT) {
            if (a.histC && lane < M.nsph && writer) a.histC[((size_t)s * a.B + traj) * M.nsph + lane] = chart[lane];
        }
    }
    if (id >= 0 && writer) {
        a.q[off] = q;
        a.qd[off] = qd;
        a.qp[off] = qp;
        a.qdp[off] = qdp;
    }
    if (LEAN && lane == 0) a.resume[traj] = stop;
    if constexpr (PARKS) {
        if (a.park && lane == 0) {
            a.resume[traj] = stop;
            if (stop < a.nsteps) {
                a.park[1 + atomicAdd(a.park, 1)] = traj;
                int* pp = a.park + 1 + a.B + 3 * traj;
                pp[0] = piv.hold; pp[1] = piv.len; pp[2] = piv.streak;
            }
        }
    }
    if (lane == 0 && a.it && writer) {
        a.it[traj] += iters;
        a.ls[traj] += halv;
        a.status[traj] |= status;
    }
    if (lane == 0 && a.ticks && writer) a.ticks[traj] += __builtin_amdgcn_s_memtime() - tick0;      // this rollout's share of the launch (rmx_step_ticks)
    }
    if constexpr (tag_w2(TAG)) w2_release(lane);
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
        const double qd1 = lu_solve_neg<NP, true>(M.n, lane, Mrow, -rhs);
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

// ---------------------------------------------------------------- adjoint BDF1 / BDF2 (BASELINE.json configs[3])
//
// taskObjective of driverRedMaxAdjointBDF1.m:39-62 (TaskBDF1PointPos) and driverRedMaxAdjointBDF2.m:38-62 (TaskBDF2PointPos), batched:
// parameters p[B][nr] are constant joint torques tau = pscale*p (applyStep, TaskBDF1PointPos.m:58-64).  Forward = simLoop (BDF1
// :65-102; BDF2 :65-136 with the SDIRK2a / SDIRK2b start step) with the line-search-free newton (:105-146 / :139-181); per step the
// H, M, D of the LAST EVALUATED iterate of the step's final solve are kept in HBM ([B][nsteps][n*n], column-major over nodes), which
// is what Scene.saveHistory stores (the reference keeps lu(H), the backward kernel re-factors H').  Backward = TaskBDF1.calcFinal
// (TaskBDF1.m:45-81) / TaskBDF2.calcFinal (TaskBDF2.m:45-107).  Every forward solve is the common residual
//     qdot = (x - qA)/eta,  v = x - qB,  g = M v - eta^2 f,  H = dg/dx      (evalBDF1, evalSDIRK2a/b, evalBDF2).

// HELP (RMX_PART 8, trees of <= 16 nodes in batches of at most one rollout per two SIMDs - configs[3]'s 512 rollouts): a workgroup of
// TWO wavefronts per rollout.  M and D of a step do not enter the Newton iteration; forming and storing them is 9.5 % of the launch
// pair (profiles/r05w_adjoint_md_bound.txt).  Wave 0 - the rollout - leaves the 39 numbers per node that eval_MD reads in a
// double-buffered hand-over area of the workgroup's LDS at the end of each step and goes on with the next step; wave 1 forms M, D
// from them and stores them.  One workgroup barrier per step: wave 1 reaches barrier s + 1 after it has finished step s, so it has
// read buffer s & 1 before wave 0 (which writes that buffer again only behind barrier s + 1) can touch it.  Same function on the same
// numbers: M, D are bit-identical to the one-wave kernel's.
constexpr int ADJ_HAND = 40;      // doubles per node and buffer
__host__ __device__ constexpr size_t adj_hand_doubles(const int NP) { return (size_t)2 * ADJ_HAND * NP; }
template <int NP>
__device__ __forceinline__ void adj_hand_put(double* __restrict__ hb, const int lane, const FrontState& fs) {
    if (lane < NP) {
        double* o = hb + lane;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            o[c * NP] = fs.sw[c];
            o[(3 + c) * NP] = fs.sv[c];
            o[(6 + c) * NP] = fs.xiw[c];
            o[(9 + c) * NP] = fs.xiv[c];
        }
#pragma unroll
        for (int c = 6; c < NACC; ++c) o[(12 + c - 6) * NP] = fs.S[c];
        static_assert(12 + NACC - 6 + 5 <= ADJ_HAND, "hand-over rows");
        o[(12 + NACC - 6) * NP] = fs.dd;
        o[(13 + NACC - 6) * NP] = __longlong_as_double((long long)fs.anc_m);
        o[(14 + NACC - 6) * NP] = __longlong_as_double((long long)fs.desc_m);
        o[(15 + NACC - 6) * NP] = fs.act ? 1.0 : 0.0;
        o[(16 + NACC - 6) * NP] = fs.dof ? 1.0 : 0.0;
    }
}
template <int NP>
__device__ __forceinline__ void adj_md_helper(const DevModel& M, const AdjArgs& a, const double* __restrict__ hand, const int lane, const int traj) {
    const int n = M.n;
    const size_t nn = (size_t)n * n;
    const int l = lane < NP ? lane : 0;      // (lanes beyond the tree's DPP row read node 0's numbers: their results are not stored)
    for (int s = 1; s <= a.nsteps; ++s) {
        __syncthreads();
        const double* o = hand + (size_t)(s & 1) * (ADJ_HAND * NP) + l;
        FrontState fs;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            fs.sw[c] = o[c * NP];
            fs.sv[c] = o[(3 + c) * NP];
            fs.xiw[c] = o[(6 + c) * NP];
            fs.xiv[c] = o[(9 + c) * NP];
        }
#pragma unroll
        for (int c = 6; c < NACC; ++c) fs.S[c] = o[(12 + c - 6) * NP];
        fs.dd = o[(12 + NACC - 6) * NP];
        fs.anc_m = (unsigned long long)__double_as_longlong(o[(13 + NACC - 6) * NP]);
        fs.desc_m = (unsigned long long)__double_as_longlong(o[(14 + NACC - 6) * NP]);
        fs.act = lane < NP && o[(15 + NACC - 6) * NP] != 0.0;
        fs.dof = o[(16 + NACC - 6) * NP] != 0.0;
        double Mrow[NP], Drow[NP];
        eval_MD<NP>(M, lane, fs, Mrow, Drow);
        double* Mk = a.Ms + ((size_t)traj * a.nsteps + (s - 1)) * nn;
        double* Dk = a.Ds + ((size_t)traj * a.nsteps + (s - 1)) * nn;
        if (lane < n) {
#pragma unroll
            for (int i = 0; i < NP; ++i)
                if (i < n) {
                    Mk[(size_t)i * n + lane] = Mrow[i];
                    Dk[(size_t)i * n + lane] = Drow[i];
                }
        }
    }
}

// FC: a serial chain that fills every node slot (is_chain && n == NP, decided by the launcher): the two model facts as compile-time
// constants (model_view), as in the FULLCHAIN step kernels - no tree paths in the front, no per-row bounds
template <int NP, int INTEG, bool HELP = false, bool FC = false>
__global__ void __launch_bounds__(HELP ? 128 : 64) k_adjoint_fwd(const DevModel Min, const DevOpts o, const AdjArgs a) {
    static_assert(!HELP || NP <= 16, "the helper-wave form: trees of one DPP row");
    const DevModel M = model_view<NP, FC>(Min);
    double *sAcc, *sCol;
    smem_setup<NP>(M, sAcc, sCol);
    double* hand = nullptr;
    if constexpr (HELP) {
        hand = sAcc + acc_doubles(M.n, NP) + (size_t)(NCONST + NGROUND) * cstride(NP);      // behind the wave's scratch and the constants
        if (threadIdx.x >= 64) {
            adj_md_helper<NP>(M, a, hand, (int)threadIdx.x - 64, (int)blockIdx.x);
            return;
        }
    }
    const int lane = threadIdx.x, traj = blockIdx.x, n = M.n;
    const int id = (lane < n) ? M.idx[lane] : -1;
    const size_t off = (size_t)traj * M.nr + (id >= 0 ? id : 0);
    double q = id >= 0 ? a.q[off] : 0.0;
    double qd = id >= 0 ? a.qd[off] : 0.0;
    double qp = 0.0, qdp = 0.0;             // BDF2: the state of step k-1 (Joint.q1 / qdot1)
    const double pj = id >= 0 ? a.p[off] : 0.0;
    const double h = o.h;
    FrontState fs;
    fs.tau_add = a.pscale * pj;
    // does the task body hang below (or at) this lane's joint?  (rows of J(idxM_body, :) that are non-zero)
    const bool on_path = lane < n && (lane == a.task_node || ((M.rel[MAXN + lane] >> a.task_node) & 1ull));
    int iters = 0, status = 0;
    double Ptask = 0.0;
    const size_t nn = (size_t)n * n;
    const double al = (2.0 - sqrt(2.0)) / 2.0;       // SDIRK2 (driverRedMaxAdjointBDF2.m:80)
    for (int s = 1; s <= a.nsteps; ++s) {
        double* Hk = a.Hs + ((size_t)traj * a.nsteps + (s - 1)) * nn;
        double* Mk = a.Ms + ((size_t)traj * a.nsteps + (s - 1)) * nn;
        double* Dk = a.Ds + ((size_t)traj * a.nsteps + (s - 1)) * nn;
        double Jw[3] = {0.0, 0.0, 0.0}, Jv[3] = {0.0, 0.0, 0.0};   // J(idxM_body, this joint) of the last evaluated iterate
        const double q0 = (INTEG == 2 && s > 1) ? qp : q, qd0 = (INTEG == 2 && s > 1) ? qdp : qd;      // BDF2: step k-1
        const double q1 = q, qd1 = qd;                                                                  // BDF2: step k
        double qa = 0.0, qda = 0.0;                  // SDIRK2a's result
        double x = 0.0, xlo = 0.0;                   // compensated iterate x + xlo (newton_impl in rmx_device.h)
        const int nsolve = (INTEG == 2 && s == 1) ? 2 : 1;
        for (int sv = 0; sv < nsolve; ++sv) {
            double qA, qB, eta;
            if (INTEG == 1) {                        // evalBDF1 :160-176
                eta = h; qA = q0; qB = q0 + h * qd0; x = qB;
            } else if (s == 1 && sv == 0) {          // SDIRK2a :78-84, evalSDIRK2a :184-216
                eta = al * h; qA = q0; qB = q0 + (al * h) * qd0; x = q0 + al * h * qd0;
            } else if (s == 1) {                     // SDIRK2b :89-92, evalSDIRK2b :219-252
                eta = al * h;
                x = qa + (1.0 - al) * h * qda;
                qA = q0 + (1.0 - al) * h * qda;
                qB = q0 + (2.0 * al - 1.0) * h * qd0 + 2.0 * (1.0 - al) * h * qda;
            } else {                                 // BDF2 :103-113, evalBDF2 :255-287
                eta = (2.0 / 3.0) * h;
                x = q1 + h * qd1;
                qA = (4.0 / 3.0) * q1 - (1.0 / 3.0) * q0;
                qB = (4.0 / 3.0) * q1 - (1.0 / 3.0) * q0 + (8.0 / 9.0) * h * qd1 - (2.0 / 9.0) * h * qd0;
            }
            xlo = 0.0;
            const bool last_solve = sv + 1 == nsolve;    // the SDIRK2a solve leaves nothing in the history
            int iter = 1;
            // Scene.saveHistory keeps H, M, D of the LAST evaluated iterate of the step (driverRedMaxAdjointBDF1.m:100, 127).  Up to 32
            // nodes H rides in registers through the Newton loop (the solve destroys its working copy) and M, D are formed ONCE, after
            // the loop, from the state the last evaluation left behind (fs) - they do not enter the Newton iteration itself; all three
            // go to HBM once per step.  Larger trees store H at every iterate (the last store wins) and form M, D once per step as well.
            constexpr bool STORE_ONCE = NP <= 32;
            double Hs[STORE_ONCE ? NP : 1];
            while (true) {
                NodeOut e;
                double Hrow[NP];
                eval_front<NP, true>(M, sAcc, lane, x, ((x - qA) + xlo) / eta, (x - qB) + xlo, eta, e, fs);
                const double hdiag = eval_hess<NP>(M, lane, fs, Hrow, nullptr, sAcc);
                if constexpr (STORE_ONCE) {
#pragma unroll
                    for (int i = 0; i < NP; ++i) Hs[i] = Hrow[i];
                } else if (last_solve) {
                    // the solve destroys its copy and 64 more doubles per lane do not fit next to it: H of every iterate goes to HBM, the
                    // last store wins.  M and D do not enter the Newton iteration: they are formed once, after the loop, from fs
                    if (lane < n) {
#pragma unroll
                        for (int i = 0; i < NP; ++i)
                            if (i < n) Hk[(size_t)i * n + lane] = Hrow[i];
                    }
                }
                if (s == a.task_step && last_solve) {   // J(body rows, joint) = Ad(E_body^-1) s_joint : body-frame twist of the task body per unit qdot
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
                // [Hl,Hu,Hp] = lu(H,'vector'); dx = -(Hu\(Hl\g(Hp)))  :127-128.  As in the step kernels the solve takes diagonal pivots
                // under the growth guard first (a third of the instructions of the pivot search) and falls back to partial pivoting
                // on the saved copy of H when the guard trips; the factors themselves are not kept (the backward pass re-solves).
                double dx;
                if constexpr (STORE_ONCE) {
                    bool lu_ok;
                    dx = lu_solve_neg_diag<NP>(lane, Hrow, e.g, hdiag, lu_ok);
                    if (!lu_ok) {
#pragma unroll
                        for (int i = 0; i < NP; ++i) Hrow[i] = Hs[i];
                        dx = lu_solve_neg<NP, true>(n, lane, Hrow, e.g);
                        status |= 16;
                    }
                } else if constexpr (NP == 64 && LU_SPLIT64) {
                    // 33..64 nodes: the guarded block-column solve of the step kernels (lu_solve_neg_diag64: H through the front's scratch,
                    // every update one DPP-fused FMA; a third of the instructions of the pivoted solve); H of this iterate is in the
                    // history already, so a tripped guard reloads it from there and pivots
                    (void)hdiag;
                    bool lu_ok;
                    dx = lu_solve_neg_diag64(n, lane, sAcc, Hrow, e.g, lu_ok);
                    if (!lu_ok) {
                        if (last_solve) {
#pragma unroll
                            for (int i = 0; i < NP; ++i) Hrow[i] = (lane < n && i < n) ? Hk[(size_t)i * n + lane] : ((i == lane) ? 1.0 : 0.0);
                        } else {             // (the SDIRK2a solve leaves nothing in the history: evaluate again)
                            NodeOut e2;
                            eval_front<NP, true>(M, sAcc, lane, x, ((x - qA) + xlo) / eta, (x - qB) + xlo, eta, e2, fs);
                            eval_hess<NP>(M, lane, fs, Hrow, nullptr, sAcc);
                        }
                        dx = lu_solve_neg<NP, true>(n, lane, Hrow, e.g);
                        status |= 16;
                    }
                } else {
                    (void)hdiag;
                    dx = lu_solve_neg<NP, true>(n, lane, Hrow, e.g);
                }
                const double dxn2 = wave_sum(dx * dx);
                if (!(dxn2 == dxn2)) { status |= 4; break; }
                if (sqrt(dxn2) > o.dxMax) { status |= 1; break; }            // :129-132
                {                                                             // x = x + dx, :134, before the convergence test
                    const double xa = x;
                    two_sum(xa, xlo + dx, x, xlo);
                    xlo *= o.comp;
                }
                if (sqrt(wave_sum(e.g * e.g)) < o.tol) break;                 // :135-138
                if (iter >= o.iterMax) { status |= 2; break; }                // :139-142
                ++iter;
            }
            if constexpr (NP == 32 && HESS_MFMA) {
                if (last_solve) {           // M, D on the matrix cores, stored from the MFMA layout (eval_MD_mfma32_store); H from its rows
                    eval_MD_mfma32_store(M, lane, fs, sAcc, Mk, Dk);
                    if (lane < n) {
#pragma unroll
                        for (int i = 0; i < NP; ++i)
                            if (i < n) Hk[(size_t)i * n + lane] = Hs[i];
                    }
                }
            } else if (HELP && last_solve) {
                adj_hand_put<NP>(hand + (size_t)(s & 1) * (ADJ_HAND * NP), lane, fs);      // fs: the state of the last evaluated iterate
                __syncthreads();                                                            // (the one barrier of the step: see above)
                if (lane < n) {
#pragma unroll
                    for (int i = 0; i < NP; ++i)
                        if (i < n) Hk[(size_t)i * n + lane] = Hs[i];
                }
            } else if (last_solve) {
                double Mrow[NP], Drow[NP];
#ifdef RMX_ADJ_SKIP_MD      // measurement aid: what the forward kernel costs without forming M, D (they are stored as zeros)
                for (int i = 0; i < NP; ++i) Mrow[i] = Drow[i] = 0.0;
#else
                eval_MD<NP>(M, lane, fs, Mrow, Drow);       // fs: the state of the last evaluated iterate
#endif
                if (lane < n) {
#pragma unroll
                    for (int i = 0; i < NP; ++i)
                        if (i < n) {
                            if constexpr (STORE_ONCE) Hk[(size_t)i * n + lane] = Hs[i];
                            Mk[(size_t)i * n + lane] = Mrow[i];
                            Dk[(size_t)i * n + lane] = Drow[i];
                        }
                }
            }
            if (INTEG == 2 && s == 1 && sv == 0) {   // :85-87
                qa = x;
                qda = ((x - q0) + xlo) / (al * h);
            }
        }
        if (INTEG == 1) {
            qd = ((x - q0) + xlo) / h;
            q = x;
        } else if (s == 1) {                         // :93-98: setQ(q1, qdot1), setQ1(q0, qdot0)
            qd = (x - q0 - (1.0 - al) * h * qda) / (al * h);
            q = x;
            qp = q0;
            qdp = qd0;
        } else {                                     // :114-117
            qd = (3.0 / (2.0 * h)) * (x - (4.0 / 3.0) * q1 + (1.0 / 3.0) * q0);
            q = x;
            qp = q1;
            qdp = qd1;
        }
        if (s == a.task_step) {    // TaskBDF1PointPos.calcStep :67-107 (TaskBDF2PointPos.calcStep is the same) at the final state of this step
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
        if (INTEG == 2) {
            a.qp[off] = qp;
            a.qdp[off] = qdp;
        }
    }
    const double preg = wave_sum(pj * pj);
    if (lane == 0) {
        a.P[traj] = Ptask + a.wreg * 0.5 * preg;      // TaskBDF1.calcFinal :49 / TaskBDF2.calcFinal :49
        if (a.it) {
            a.it[traj] = iters;
            a.status[traj] = status;
        }
    }
}

// yk -= (cm M_j + cd h D_j)' z_j : one off-diagonal block of the backward sweep; this lane's entry = column `col` of the block . z
template <int NP>
__device__ __forceinline__ void adj_block(double& y, const double* __restrict__ Mj, const double* __restrict__ Dj, const int n, const int lane,
                                          const int col, const double cm, const double cdh, const double z) {
    const double* Mc = Mj + (size_t)col * n;
    const double* Dc = Dj + (size_t)col * n;
    // Every load of the column before its first use, none of them under a lane condition (index and column clamped into the block,
    // the padding selected afterwards): guarded by `j < n && lane < n` each load sat in its own exec-masked block with a full wait
    // behind it - 32 dependent round trips per block and step, most of the backward kernel's time.
    double mc[NP], dc[NP];
    const bool withD = cdh != 0.0;
#pragma unroll
    for (int j = 0; j < NP; ++j) mc[j] = Mc[j < n ? j : 0];
    if (withD) {
#pragma unroll
        for (int j = 0; j < NP; ++j) dc[j] = Dc[j < n ? j : 0];
    } else {
#pragma unroll
        for (int j = 0; j < NP; ++j) dc[j] = 0.0;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        double blk = 0.0;
        if (j < n && lane < n) blk = withD ? (cm * mc[j] + cdh * dc[j]) : cm * mc[j];
        y -= blk * readlane_d(z, j);
    }
}

#ifndef RMX_ADJ_BWD_PIVOT
#define RMX_ADJ_BWD_PIVOT 0          // 1: the backward sweep's solves with the pivot search always (build variants)
#endif
template <int NP, int INTEG, bool FC = false>
__global__ void __launch_bounds__(64) k_adjoint_bwd(const DevModel Min, const DevOpts o, const AdjArgs a) {
    const DevModel M = model_view<NP, FC>(Min);
    const int lane = threadIdx.x, traj = blockIdx.x, n = M.n;
    const int id = (lane < n) ? M.idx[lane] : -1;
    const size_t nn = (size_t)n * n;
    const double h = o.h;
    const int col = lane < n ? lane : 0;
    const double al = (2.0 - sqrt(2.0)) / 2.0;
    double z1 = 0.0, z2 = 0.0, z3 = 0.0, z4 = 0.0, zs = 0.0;      // z of steps k+1 .. k+4
    const double* Mb = a.Ms + (size_t)traj * a.nsteps * nn;
    const double* Db = a.Ds + (size_t)traj * a.nsteps * nn;
    for (int k = a.nsteps; k >= 1; --k) {
        double y = (k == a.task_step && lane < n) ? a.dPdq[(size_t)traj * n + lane] : 0.0;
        if (INTEG == 1) {
            // yk -= (-2 M_{k+1} + h D_{k+1})' z_{k+1}   TaskBDF1.m:58-64 ;   yk -= M_{k+2}' z_{k+2}   :65-70
            if (k < a.nsteps) adj_block<NP>(y, Mb + (size_t)k * nn, Db + (size_t)k * nn, n, lane, col, -2.0, h, z1);
            if (k < a.nsteps - 1) adj_block<NP>(y, Mb + (size_t)(k + 1) * nn, Db + (size_t)(k + 1) * nn, n, lane, col, 1.0, 0.0, z2);
        } else {
            // TaskBDF2.m:66-96: blocks of steps k+1 .. k+4; the k == 1 variants carry the SDIRK2 start step's coefficients
            if (k < a.nsteps)
                adj_block<NP>(y, Mb + (size_t)k * nn, Db + (size_t)k * nn, n, lane, col, k == 1 ? -((8.0 / (9.0 * al)) + (4.0 / 3.0)) : -(8.0 / 3.0),
                              (8.0 / 9.0) * h, z1);
            if (k < a.nsteps - 1)
                adj_block<NP>(y, Mb + (size_t)(k + 1) * nn, Db + (size_t)(k + 1) * nn, n, lane, col, k == 1 ? ((2.0 / (9.0 * al)) + (19.0 / 9.0)) : (22.0 / 9.0),
                              -(2.0 / 9.0) * h, z2);
            if (k < a.nsteps - 2) adj_block<NP>(y, Mb + (size_t)(k + 2) * nn, Db + (size_t)(k + 2) * nn, n, lane, col, -(8.0 / 9.0), 0.0, z3);
            if (k < a.nsteps - 3) adj_block<NP>(y, Mb + (size_t)(k + 3) * nn, Db + (size_t)(k + 3) * nn, n, lane, col, 1.0 / 9.0, 0.0, z4);
        }
        // z_k = H_k'^-1 y_k  (zkk0(Hp) = Hl'\(Hu'\yk) :76): this lane's "row" of H' is column `lane` of H
        double Hrow[NP];
        const double* Hc = a.Hs + ((size_t)traj * a.nsteps + (k - 1)) * nn + (size_t)col * n;
        {   // (unconditional loads, all in flight together: see adj_block)
            double hv[NP];
#pragma unroll
            for (int i = 0; i < NP; ++i) hv[i] = Hc[i < n ? i : 0];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NP; ++i) Hrow[i] = (i < n && lane < n) ? hv[i] : ((i == lane) ? 1.0 : 0.0);
        }
        double z;
        if constexpr (NP <= 32 && !RMX_ADJ_BWD_PIVOT) {
            // as in the forward sweep: diagonal pivots under the growth guard first (a third of the instructions of the pivot search; H'
            // is as close to symmetric positive definite as H), partial pivoting on a fresh copy of the rows when the guard trips
            const double hdl = Hc[col];
            const double hd = lane < n ? hdl : 1.0;
            bool lu_ok;
            z = lu_solve_neg_diag<NP>(lane, Hrow, -y, hd, lu_ok);
            if (!lu_ok) {
#pragma unroll
                for (int i = 0; i < NP; ++i) Hrow[i] = (i < n && lane < n) ? Hc[i] : ((i == lane) ? 1.0 : 0.0);
                z = lu_solve_neg<NP, true>(n, lane, Hrow, -y);
            }
        } else {
            z = lu_solve_neg<NP, true>(n, lane, Hrow, -y);
        }
        zs += z;
        z4 = z3;
        z3 = z2;
        z2 = z1;
        z1 = z;
    }
    if (id >= 0) {   // dPdp = wreg*p' - z'*dgdp, dgdp(kk,:) = -eta^2*pscale*I with eta^2 = h^2 (TaskBDF1PointPos.m:104-105) or (4/9) h^2 for
        const size_t off = (size_t)traj * M.nr + id;                 // EVERY step (TaskBDF2PointPos.m:97-106)
        const double e2 = INTEG == 1 ? h * h : (4.0 / 9.0) * h * h;
        a.dPdp[off] = a.wreg * a.p[off] + e2 * a.pscale * zs;
    }
}

// Parity hook: one residual (+Hessian) evaluation per trajectory, results to HBM.
template <int NP, bool WANT_H, bool CT>
__global__ void __launch_bounds__(64) k_eval(const DevModel M, const int B, const double* __restrict__ q,
                                             const double* __restrict__ qA, const double* __restrict__ qB, const double eta,
                                             double* __restrict__ g, double* __restrict__ H, const int* __restrict__ chart) {
    double *sAcc, *sCol;
    smem_setup<NP>(M, sAcc, sCol);
    const int lane = threadIdx.x, traj = blockIdx.x;
    if constexpr (CT) {
        con_setup<NP>(M, sCol);
        if (M.nsph) sph_setup<NP>(M, sCol, lane, chart + (size_t)traj * M.nsph);
    }
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

// Parity hook for computeValues itself (driverRedMaxBDF1.m:190-243): M = J'MmJ (:212), f = fr + J'(fm - Mm Jdot qdot) (:215-216)
// and D = df/dqdot (:227-237) at (q, qdot), results to HBM (column-major per trajectory).  f is the residual with v = 0 and
// e2 = 1 (g = M v - e2 f = -f); M and D rows come from the subtree sums the same front pass leaves behind (eval_MD).
template <int NP, bool CT = false>
__global__ void __launch_bounds__(64) k_eval_mfd(const DevModel M, const int B, const double* __restrict__ q, const double* __restrict__ qd,
                                                 double* __restrict__ Mo, double* __restrict__ fo, double* __restrict__ Do,
                                                 const int* __restrict__ chart) {
    double *sAcc, *sCol;
    smem_setup<NP>(M, sAcc, sCol);
    // JointSpherical / JointFree3D: the group's three revolute nodes about the axes of the trajectory's Euler chart ARE the joint in
    // the chart's coordinates (S = T of JointSpherical.m:298-303), so M, f, D come out in those coordinates with no further term
    if constexpr (CT) con_setup<NP>(M, sCol);
    if (M.nsph) sph_setup<NP>(M, sCol, threadIdx.x, chart + (size_t)blockIdx.x * M.nsph);
    const int lane = threadIdx.x, traj = blockIdx.x;
    const int id = (lane < M.n) ? M.idx[lane] : -1;
    const size_t off = (size_t)traj * M.nr + (id >= 0 ? id : 0);
    FrontState fs;
    NodeOut e;
    eval_front_e2<NP, true, false, CT>(M, sAcc, lane, id >= 0 ? q[off] : 0.0, id >= 0 ? qd[off] : 0.0, 0.0, 1.0, 1.0, e, fs);
    double Mrow[NP], Drow[NP];
    if constexpr (CT) {       // ForceGroundCuboid: its wrench is in f (the front), its damping blocks go into D here
        double yc[6];
        contact_damping_fold<NP>(M, sAcc, lane, fs, yc);
        eval_MD<NP, true>(M, lane, fs, Mrow, Drow, yc);
    } else {
        eval_MD<NP>(M, lane, fs, Mrow, Drow);
    }
    if (id >= 0) fo[off] = -e.g;
    double* Mt = Mo + (size_t)traj * M.nr * M.nr;
    double* Dt = Do + (size_t)traj * M.nr * M.nr;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        if (i < M.n) {
            const int ci = M.idx[i];
            if (id >= 0 && ci >= 0) {
                Mt[(size_t)ci * M.nr + id] = Mrow[i];
                Dt[(size_t)ci * M.nr + id] = Drow[i];
            }
        }
    }
}

// Joint.computeEnergies / Body.computeEnergies at the stored state.
template <int NP, bool CT>
__global__ void __launch_bounds__(64) k_energy(const DevModel M, const int B, const double* __restrict__ q,
                                               const double* __restrict__ qd, double* __restrict__ T, double* __restrict__ V,
                                               const int* __restrict__ chart) {
    double *sAcc, *sCol;
    smem_setup<NP>(M, sAcc, sCol);
    if constexpr (CT) {
        con_setup<NP>(M, sCol);
        if (M.nsph) sph_setup<NP>(M, sCol, threadIdx.x, chart + (size_t)blockIdx.x * M.nsph);
    }
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

// Profiling hook: shader-clock cycles (s_memtime) of the phases of one Newton iteration with the GENERIC device functions (eval_node,
// the pivoting solve / the 64-row guarded solve) at the production occupancy (one wavefront per trajectory).  The full 32-link chain
// is timed by k_phase_time_pair32 instead (the functions its step kernel runs).
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
        bool staged = false;
        if constexpr (NP == 64 && LU_SPLIT64 && HESS_MFMA64) {
            if (M.tree_dmax > 0) {                    // a branching tree: the Hessian stage as the step kernels run it (operands staged for tree_solve64)
                FrontState fs;
                eval_front<NP, true, true, false>(M, sAcc, lane, x, (x - q0) / h, x - (q0 + h * qd0), h, e, fs, stamps);
                (void)eval_hess<NP, false, false, false>(M, lane, fs, Hrow, nullptr, sAcc, e.g);
                staged = true;
            }
        }
        if (!staged) eval_node<NP, true, true>(M, sAcc, sCol, lane, x, (x - q0) / h, x - (q0 + h * qd0), h, e, Hrow, stamps);
        unsigned long long t2 = __builtin_amdgcn_s_memtime();
        double dx;
        if constexpr (NP == 64 && LU_SPLIT64) {      // the guarded solve the step kernels run (33..64 rows)
            bool lu_ok;
            if (M.tree_dmax > 0) {                    // a branching tree: along the tree
                dx = solve64_staged(M, lane, sAcc, lu_ok);
            } else {
                dx = lu_solve_neg_diag64(M.n, lane, sAcc, Hrow, e.g, lu_ok);
            }
            sink += lu_ok ? 0.0 : 1.0;
        } else {
            dx = lu_solve_neg<NP, true>(M.n, lane, Hrow, e.g);
        }
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

// ============================================================================ launchers (declared in rmx_host.h)
//
// RMX_PART 0: the plain kernels (every scene without ForceGroundCuboid / JointSpherical) plus Euler, adjoint, phase timing.
// RMX_PART 1: the extended (CT) instantiations of eval / step / energy.
// RMX_PART 2: the FULLCHAIN instantiations of the plain step kernels.  One object per part and size, so the builds run in parallel.
#ifndef RMX_PART
#define RMX_PART 0
#endif

// More than 64 KiB of dynamic LDS (trees of 33..64 nodes: 34 KiB of per-node constants + H row-major for the block-column solve)
// is an opt-in per kernel and device; it costs microseconds, so it is set at every launch rather than cached (a process may
// drive several devices from several threads).
#define RMX_STR_(x) #x
#define RMX_STR(x) RMX_STR_(x)
#define RMX_LAUNCH(kernel, grid, block, bytes, stream, ...)                                                                         \
    do {                                                                                                                            \
        if ((bytes) > 64 * 1024)                                                                                                    \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)); \
        kernel<<<grid, block, bytes, stream>>>(__VA_ARGS__);                                                                        \
    } while (0)

#if RMX_PART == 3      // 64-lane plain step kernels with the per-node constants in global memory (RMX_GLOBAL_CONSTS): four wavefronts per CU
#ifndef RMX_GLOBAL_CONSTS
#error "RMX_PART 3 is compiled with -DRMX_GLOBAL_CONSTS"
#endif

void RMX_CAT(launch_step_gconst_, RMX_NP)(const rmx_model* m, const rmx_batch* b, int integ, const DevOpts& o, const StepArgs& a) {
    const dim3 grid(b->B), block(64);
    b->last_kernel = integ == INTEG_BDF1 ? "k_step_bdf1<64,gconst>" : "k_step_bdf2<64,gconst>";
    const size_t bytes = sizeof(double) * (size_t)acc_doubles(m->n, RMX_NP);
    if (m->dm.n == RMX_NP) {          // every node slot in use: the n == NP instantiation
        if (integ == INTEG_BDF1) RMX_LAUNCH((k_step_bdf1<RMX_NP, false, false, false, TAG_FULLN + 1>), grid, block, bytes, b->stream, m->dm, o, a);
        else RMX_LAUNCH((k_step_bdf2<RMX_NP, false, false, false, TAG_FULLN + 1>), grid, block, bytes, b->stream, m->dm, o, a);
        return;
    }
    if (integ == INTEG_BDF1) RMX_LAUNCH((k_step_bdf1<RMX_NP, false, false, false, 3>), grid, block, bytes, b->stream, m->dm, o, a);
    else RMX_LAUNCH((k_step_bdf2<RMX_NP, false, false, false, 3>), grid, block, bytes, b->stream, m->dm, o, a);
}

#elif RMX_PART == 4      // serial chains of <= 32 nodes with ForceGroundCuboid: the step kernels around newton_pair (rmx_ct32.h)
#if RMX_NP != 32
#error "RMX_PART 4 is compiled for RMX_NP = 32"
#endif
#include "rmx_ct32.h"

// simLoop of driverRedMaxBDF1.m:57-91 / driverRedMaxBDF2.m:57-125 (integ, wave-uniform) for one rollout of a chain of <= 32 nodes with
// ground contact, steps sfirst .. nsteps - 1.  ONE call site of newton_pair for every stage of every integrator: the SDIRK2 start step
// is two passes of the stage loop, everything else one.
// Returns the step at which the rollout was handed on (nsteps: it is complete).  One Newton flavour per instantiation:
// RUN_LEAN: free flight - the lean solve (newton_node<32, true, true>: the plain evaluation plus the test that every cuboid is clear of
//   the ground, under which the contact terms vanish identically); the rollout is handed on at the start of the first STEP in which an
//   evaluation fails that test (what the lean launch of launch_step_ct_32 does).
// RUN_PAIR: newton_pair, one wavefront; a solve whose line searches keep running out their trials hands the rollout on, at the start
//   of that step, to a cooperative group (DevOpts::parkHalv).
// RUN_COOP: this wavefront is a member of the group that finishes a parked rollout.
enum { RUN_LEAN = 0, RUN_PAIR = 1, RUN_COOP = 2 };
template <int MODE>
__device__ __forceinline__ int run_rollout(const DevModel& M, const DevOpts& o, const StepArgs& a, const int integ, double* sAcc, double* sCol,
                                           const int lane, const int traj, const int sfirst, CoopCtx& cx, const CoopPub& pb,
                                           const unsigned long long tick0) {
    constexpr int NP = 32;
    constexpr bool COOP = MODE == RUN_COOP;
    const bool writer = !COOP || cx.member == 0;     // (members 1.. of a cooperative group compute, member 0 also stores)
    const double h = o.h;
    const bool bdf2 = integ == INTEG_BDF2;
    const int id = (lane < M.n) ? M.idx[lane] : -1;
    const size_t off = (size_t)traj * M.nr + (id >= 0 ? id : 0);
    double q = id >= 0 ? a.q[off] : 0.0;
    double qd = id >= 0 ? a.qd[off] : 0.0;
    double qp = (bdf2 && id >= 0) ? a.qp[off] : 0.0;       // step k-1 (Joint.q1 / qdot1 in the reference)
    double qdp = (bdf2 && id >= 0) ? a.qdp[off] : 0.0;
    const bool started = (*a.started) != 0 || sfirst > 0;    // resumed behind earlier steps of this call: they are its history
    int iters = 0, halv = 0, status = 0;
    PivotPolicy piv;
    if constexpr (COOP) {
        const int* pp = a.park + 1 + a.B + 3 * traj;
        piv.hold = pp[0]; piv.len = pp[1]; piv.streak = pp[2];
    }
    int stop = a.nsteps;
    for (int s = sfirst; s < a.nsteps; ++s) {
        NodeOut last;
        last.g = last.eT = last.eV = 0.0;
        double xlo = 0.0;
        const int it_in = iters, hv_in = halv, st_in = status;
        const PivotPolicy piv_in = piv;
        const bool start2 = bdf2 && s == 0 && !started;       // SDIRK2 start step (driverRedMaxBDF2.m:64-88): two solves
        const double al = (2.0 - sqrt(2.0)) / 2.0;            // (:74)
        const double q0 = (bdf2 && !start2) ? qp : q, qd0 = (bdf2 && !start2) ? qdp : qd, q1 = q, qd1 = qd;
        double qa = 0.0, qda = 0.0, xsol = 0.0;
        bool left = false;
        for (int stage = 0; stage < (start2 ? 2 : 1) && !left; ++stage) {
            double xi, qA, qB, eta;
            if (!bdf2) {                       // BDF1 (evalBDF1 :160-187): eta = h, qA = q0, qB = q0 + h qdot0 = the initial guess (:70)
                xi = q0 + h * qd0; qA = q0; qB = xi; eta = h;
            } else if (!start2) {              // BDF2 (evalBDF2 :263-293): eta = 2h/3
                xi = q1 + h * qd1;
                qA = (4.0 / 3.0) * q1 - (1.0 / 3.0) * q0;
                qB = (4.0 / 3.0) * q1 - (1.0 / 3.0) * q0 + (8.0 / 9.0) * h * qd1 - (2.0 / 9.0) * h * qd0;
                eta = (2.0 / 3.0) * h;
            } else if (stage == 0) {           // SDIRK2a (evalSDIRK2a :194-225): eta = a h, qA = q0, qB = q0 + a h qdot0
                xi = q0 + al * h * qd0; qA = q0; qB = q0 + (al * h) * qd0; eta = al * h;
            } else {                           // SDIRK2b (evalSDIRK2b :228-260)
                xi = qa + (1.0 - al) * h * qda;
                qA = q0 + (1.0 - al) * h * qda;
                qB = q0 + (2.0 * al - 1.0) * h * qd0 + 2.0 * (1.0 - al) * h * qda;
                eta = al * h;
            }
            if constexpr (MODE == RUN_LEAN) {
                xsol = newton_node<NP, true, true, false>(M, o, sAcc, sCol, lane, xi, qA, qB, eta, last, iters, halv, status, piv, xlo, cx);
                left = (status & ST_LEFT_LEAN) != 0;   // a cuboid comes near the ground: nothing of this step is kept
            } else {
                // the pivot policy of newton_policy (rmx_device.h): a hold after three tripped solves in a row
                const bool pivot_all = o.lu_mode != 0 || piv.hold > 0;
                if (piv.hold > 0) --piv.hold;
                xsol = newton_pair<COOP>(M, o, sAcc, lane, xi, qA, qB, eta, last, iters, halv, status, piv, pivot_all, xlo, cx, pb);
                if (!pivot_all) pivot_policy_update(piv);
                left = (MODE == RUN_PAIR && (status & ST_PARK)) || (COOP && (status & ST_COOP_FAULT));
            }
            if (start2 && stage == 0 && !left) {
                qa = xsol;
                qda = (qa - q0) / (al * h);
            }
        }
        if (left) {
            if (COOP) break;                   // (ST_COOP_FAULT stays in the status)
            // nothing of this step is kept: whoever takes the rollout on starts the step again (a solve of the start step that went through included)
            iters = it_in; halv = hv_in; status = st_in & ~(ST_PARK | ST_LEFT_LEAN); piv = piv_in;
            stop = s;
            break;
        }
        if (!bdf2) {
            qd = ((xsol - q0) + xlo) / h;      // (:72), with the low-order part of the iterate the residual was evaluated at
            q = xsol;
        } else if (start2) {
            qd = (xsol - q0 - (1.0 - al) * h * qda) / (al * h);
            q = xsol;
            qp = q0;
            qdp = qd0;
        } else {
            qp = q1;
            qdp = qd1;
            qd = (3.0 / (2.0 * h)) * (xsol - (4.0 / 3.0) * q1 + (1.0 / 3.0) * q0);
            q = xsol;
        }
        if (a.histT && writer) {               // Scene.saveHistory (Scene.m:134-161)
            const double T = wave_sum(last.eT), V = wave_sum(last.eV);
            if (lane == 0) {
                a.histT[(size_t)s * a.B + traj] = T;
                a.histV[(size_t)s * a.B + traj] = V;
            }
        }
        if (a.histQ && id >= 0 && writer) {
            a.histQ[(size_t)s * a.B * M.nr + off] = q;
            a.histQd[(size_t)s * a.B * M.nr + off] = qd;
        }
    }
    if (id >= 0 && writer) {
        a.q[off] = q;
        a.qd[off] = qd;
        if (bdf2) {
            a.qp[off] = qp;
            a.qdp[off] = qdp;
        }
    }
    if constexpr (COOP) {
        // a parked rollout a group has taken to its end: k_park_audit tells it from one nobody picked up by this
        if (lane == 0 && writer && !(status & ST_COOP_FAULT)) a.resume[traj] = a.nsteps;
    }
    if constexpr (!COOP) {
        if (lane == 0) {
            a.resume[traj] = stop;
            if (MODE == RUN_PAIR && a.park && stop < a.nsteps) {
                int* pp = a.park + 1 + a.B + 3 * traj;
                pp[0] = piv.hold; pp[1] = piv.len; pp[2] = piv.streak;
            }
        }
    }
    if (lane == 0 && a.it && writer) {
        a.it[traj] += iters;
        a.ls[traj] += halv;
        a.status[traj] |= status;
    }
#ifdef RMX_TICK_PHASE
    if (lane == 0 && a.ticks && writer) a.ticks[traj] += cx.phase;
    cx.phase = 0;
#else
    if (lane == 0 && a.ticks && writer) a.ticks[traj] += __builtin_amdgcn_s_memtime() - tick0;      // this rollout's share of the launch (rmx_step_ticks)
#endif
    return stop;
}

// Three launches (RMX_GROUND_FUSED=0, and whenever the cooperative groups are switched off): the lean launch of launch_step_ct_32, then
// COOP = false for every rollout from its a.resume, then COOP = true: group g finishes the parked rollouts g, g + ngroups, ...
template <bool COOP>
__global__ void __launch_bounds__(64) k_step_pair(const DevModel M, const DevOpts o, const StepArgs a, const int integ) {
    constexpr int NP = 32;
    unsigned long long tick0 = __builtin_amdgcn_s_memtime();
    int traj = blockIdx.x;
    CoopCtx cx;
    cx.ticks = o.coopTicks;
    CoopPub pb;
    int pk = 0, npark = 1, pstride = 1;
    if constexpr (COOP) {
#ifdef RMX_COOP_MAP_AID      // measurement builds: RMX_COOP_MAP=1 scatters the members of a group over the launch (member-major mapping)
        pk = a.coop_map ? blockIdx.x % a.ngroups : blockIdx.x / COOP_G;
        cx.member = a.coop_map ? blockIdx.x / a.ngroups : blockIdx.x % COOP_G;
#else
        pk = blockIdx.x / COOP_G;
        cx.member = blockIdx.x % COOP_G;
#endif
        cx.words = a.xch + (size_t)pk * COOP_WORDS;
        pb.rec = a.xrec + (size_t)pk * 2 * COOP_REC;
        npark = a.park[0];
        pstride = a.ngroups;
        if (pk >= npark) return;
        traj = a.park[1 + pk] - 1;
    }
    const int s0 = a.resume ? a.resume[traj] : 0;
    if (!COOP && s0 >= a.nsteps) return;           // the lean launch took this trajectory all the way
    double *sAcc, *sCol;
    smem_setup<NP>(M, sAcc, sCol);
    const int lane = threadIdx.x;
    con_setup<NP>(M, sCol);
    for (; pk < npark; pk += pstride) {              // (one pass unless COOP)
        int sfirst = s0;
        if constexpr (COOP) {
            traj = a.park[1 + pk] - 1;
            sfirst = a.resume[traj];
            tick0 = __builtin_amdgcn_s_memtime();
        }
        const int stop = run_rollout<COOP ? RUN_COOP : RUN_PAIR>(M, o, a, integ, sAcc, sCol, lane, traj, sfirst, cx, pb, tick0);
        if constexpr (!COOP) {
            if (lane == 0 && a.park && stop < a.nsteps) a.park[1 + atomicAdd(a.park, 1)] = traj + 1;
        }
    }
}

// ONE launch for the whole call: workgroups 0 .. B - 1 are the rollouts (lean solve, then newton_pair from the step that comes near the
// ground, until the end or until a solve parks the rollout), workgroups B .. are the members of the cooperative groups - workgroups are
// dispatched in index order (per XCD), so they take the SIMDs that finished rollouts leave - and pick the parked rollouts up as they
// appear: group g the g-th, (g + ngroups)-th ... entry of the list.  No launch boundary anywhere: a rollout that leaves free flight
// early is not held back by the last one to do so, and a parked rollout does not wait for the last unparked one.
// The list: a.park[1 + e] = rollout + 1 (zero before the launch), published with release semantics after the rollout's state;
// a.park[1 + 4 B] counts the rollout workgroups that have finished (the groups leave when all have and the list is exhausted).
// The three roles are OUT-OF-LINE functions: inlined into one kernel their three Newton loops share one register allocation (688 bytes
// of scratch, 860 spilled SGPRs, every loop slower than in a kernel of its own).  They take the launch's arguments as a pointer
// into global memory (scalar loads, as kernel arguments are) and name the LDS array themselves: a generic pointer into LDS handed
// to an out-of-line function loses its address space.
struct GroundArgs {
    DevModel M;
    DevOpts o;
    StepArgs a;
    int integ;
    int coop_only;      // measurement aid (RMX_GROUND_FUSED=3): every workgroup of this launch is a member of a cooperative group
};
__device__ __forceinline__ void role_smem(const DevModel& M, double*& sAcc, double*& sCol) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    sAcc = smem;
    sCol = smem + acc_doubles(M.n, 32);
}
// (the arguments are copied into locals once: read through the pointer, every field would be loaded again behind every store and
// every scheduling pin of the Newton loop - the compiler cannot know that nothing writes them)
__device__ __attribute__((noinline)) int role_lean(const GroundArgs* __restrict__ g, const int traj) {
    const DevModel M = g->M;
    const DevOpts o = g->o;
    const StepArgs a = g->a;
    const int integ = g->integ;
    double *sAcc, *sCol;
    role_smem(M, sAcc, sCol);
    CoopCtx cx;
    cx.ticks = o.coopTicks;
    CoopPub pb;
    return run_rollout<RUN_LEAN>(M, o, a, integ, sAcc, sCol, threadIdx.x, traj, 0, cx, pb, __builtin_amdgcn_s_memtime());
}
__device__ __attribute__((noinline)) int role_pair(const GroundArgs* __restrict__ g, const int traj, const int sfirst) {
    const DevModel M = g->M;
    const DevOpts o = g->o;
    const StepArgs a = g->a;
    const int integ = g->integ;
    double *sAcc, *sCol;
    role_smem(M, sAcc, sCol);
    CoopCtx cx;
    cx.ticks = o.coopTicks;
    CoopPub pb;
    return run_rollout<RUN_PAIR>(M, o, a, integ, sAcc, sCol, threadIdx.x, traj, sfirst, cx, pb, __builtin_amdgcn_s_memtime());
}
__device__ __forceinline__ void role_coop(const GroundArgs* __restrict__ g, const int grp, const int member) {
    const DevModel M = g->M;
    const DevOpts o = g->o;
    const StepArgs a = g->a;
    const int integ = g->integ;
    double *sAcc, *sCol;
    role_smem(M, sAcc, sCol);
    const int lane = threadIdx.x;
    CoopCtx cx;
    cx.ticks = o.coopTicks;
    CoopPub pb;
    cx.member = member;
    cx.words = a.xch + (size_t)grp * COOP_WORDS;
    pb.rec = a.xrec + (size_t)grp * 2 * COOP_REC;
    int* const done = a.park + 1 + 4 * a.B;
    for (int e = grp; e < a.B; e += a.ngroups) {
        int v = 0;
        // (the wait below ends when the rollout workgroups have all finished or parked.  It relies on their being dispatched - nothing
        // in the programming model promises that workgroups start in index order -, so it is bounded: a group that has seen no entry and
        // no end for 16 x the group timeout leaves, and k_park_audit marks whatever stays unfinished RMX_ST_COOP_FAULT)
        const unsigned long long tw0 = __builtin_amdgcn_s_memtime();
        while (true) {
            // (relaxed polls: an agent-scope ACQUIRE invalidates this XCD's L2 under every wavefront that lives in it, hundreds of
            // times per microsecond with ~500 idle members polling; the one fence below, after the entry has been seen, is what orders
            // the reads of the rollout's state)
            if (lane == 0) v = __hip_atomic_load(a.park + 1 + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v = __builtin_amdgcn_readfirstlane(v);
            if (v != 0) break;
            int d = 0, cnt = 0;
            if (lane == 0) {
                d = __hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                cnt = __hip_atomic_load(a.park, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            d = __builtin_amdgcn_readfirstlane(d);
            cnt = __builtin_amdgcn_readfirstlane(cnt);
            // every rollout has finished or parked (its list entry is written BEFORE it is counted, both by the same lane with release
            // semantics), and the count of entries, read after the count of finished rollouts, does not reach this one
            if (d >= a.B && cnt <= e) return;
            if (__builtin_amdgcn_s_memtime() - tw0 > 16ull * cx.ticks) return;
            __builtin_amdgcn_s_sleep(127);
        }
        __threadfence();                                 // acquire
        const int traj = v - 1;
        const int sfirst = __hip_atomic_load(a.resume + traj, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        run_rollout<RUN_COOP>(M, o, a, integ, sAcc, sCol, lane, traj, sfirst, cx, pb, __builtin_amdgcn_s_memtime());
    }
}
__global__ void __launch_bounds__(64) k_ground32(const GroundArgs* __restrict__ g) {
    constexpr int NP = 32;
    {
        double *sAcc, *sCol;
        smem_setup<NP>(g->M, sAcc, sCol);
        con_setup<NP>(g->M, sCol);
    }
    const int B = g->coop_only ? 0 : g->a.B, nsteps = g->a.nsteps;
    if ((int)blockIdx.x >= B) {
        role_coop(g, ((int)blockIdx.x - B) / COOP_G, ((int)blockIdx.x - B) % COOP_G);
        return;
    }
    const int traj = blockIdx.x;
    int stop = role_lean(g, traj);
    if (stop < nsteps) stop = role_pair(g, traj, stop);
    int* const park = g->a.park;
    if (!park) return;                                   // (no cooperative groups in this call: nothing parks, nobody waits)
    __threadfence();                                     // release: this rollout's state, counters and pivot policy before its list entry
    if (threadIdx.x == 0) {
        if (stop < nsteps) {
            const int e = atomicAdd(park, 1);
            __hip_atomic_store(park + 1 + e, traj + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        __hip_atomic_fetch_add(park + 1 + 4 * B, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// the steps with the contact terms of a chain of <= 32 nodes: fused (one launch for everything) or behind the lean launch of launch_step_ct_32
// After the launches of a call that may park rollouts: a rollout that was parked and that no group took to its end (a group that gave
// up waiting, see role_coop; never observed) must not pass for a result - RMX_ST_COOP_FAULT | RMX_ST_NAN and a NaN state, as for a rollout
// whose group faulted.
__global__ void __launch_bounds__(256) k_park_audit(const StepArgs a, const int nr) {
    const int traj = blockIdx.x * blockDim.x + threadIdx.x;
    if (traj >= a.B || a.resume[traj] >= a.nsteps) return;
    if (a.status) a.status[traj] |= ST_COOP_FAULT | 4;
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    for (int i = 0; i < nr; ++i) {
        a.q[(size_t)traj * nr + i] = nan;
        a.qd[(size_t)traj * nr + i] = nan;
    }
}
static void launch_step_pair_32_impl(const rmx_model* m, const rmx_batch* b, int integ, const DevOpts& o, const StepArgs& a, bool fused);
void launch_step_pair_32(const rmx_model* m, const rmx_batch* b, int integ, const DevOpts& o, const StepArgs& a, bool fused) {
    launch_step_pair_32_impl(m, b, integ, o, a, fused);
    if (a.park && o.parkHalv > 0) k_park_audit<<<dim3((b->B + 255) / 256), dim3(256), 0, b->stream>>>(a, m->nr);
}
static void launch_step_pair_32_impl(const rmx_model* m, const rmx_batch* b, int integ, const DevOpts& o, const StepArgs& a, bool fused) {
    b->last_kernel = fused ? "k_ground32" : "k_step_pair";
    if (fused) {
        // a.fused 1: rollouts and cooperative groups in one launch; 2: the rollouts (free flight + contact terms) in one launch, the groups in a second
        const int inline_groups = a.fused == 1 ? a.ngroups : 0;
        GroundArgs ga;
        ga.M = m->dm; ga.o = o; ga.a = a; ga.integ = integ;
        ga.a.ngroups = inline_groups;
        ga.coop_only = 0;
        static_assert(2 * sizeof(GroundArgs) <= RMX_GARGS_BYTES, "rmx_batch::gargs");
        // (pageable source: staged before the call returns; a failed copy must not be followed by a launch that reads the block - the
        // sticky error surfaces in the caller's hipGetLastError)
        if (hipMemcpyAsync(b->gargs, &ga, sizeof ga, hipMemcpyHostToDevice, b->stream) != hipSuccess) return;
        RMX_LAUNCH(k_ground32, dim3(b->B + inline_groups * COOP_G), dim3(64), m->smem_bytes, b->stream, (const GroundArgs*)b->gargs);
        if (a.fused == 3 && a.park && o.parkHalv > 0) {      // measurement aid: the groups as a second launch of the SAME kernel (its out-of-line role)
            ga.a.ngroups = a.ngroups;
            ga.coop_only = 1;
            GroundArgs* g2 = (GroundArgs*)b->gargs + 1;
            if (hipMemcpyAsync(g2, &ga, sizeof ga, hipMemcpyHostToDevice, b->stream) != hipSuccess) return;
            RMX_LAUNCH(k_ground32, dim3(a.ngroups * COOP_G), dim3(64), m->smem_bytes, b->stream, (const GroundArgs*)g2);
            return;
        }
        if (a.fused != 1 && a.park && o.parkHalv > 0)
            RMX_LAUNCH((k_step_pair<true>), dim3(a.ngroups * COOP_G), dim3(64), m->smem_bytes, b->stream, m->dm, o, a, integ);
        return;
    }
    RMX_LAUNCH((k_step_pair<false>), dim3(b->B), dim3(64), m->smem_bytes, b->stream, m->dm, o, a, integ);
    // group g = workgroups COOP_G g .. COOP_G g + COOP_G - 1, all of them resident at once
    if (a.park && o.parkHalv > 0) RMX_LAUNCH((k_step_pair<true>), dim3(a.ngroups * COOP_G), dim3(64), m->smem_bytes, b->stream, m->dm, o, a, integ);
}

#elif RMX_PART == 5      // 64-node trees, two wavefronts per rollout (RMX_W2): batches of up to one rollout per two SIMDs
#if RMX_NP != 64 || !RMX_W2
#error "RMX_PART 5 is compiled for RMX_NP = 64 with -DRMX_W2=1 and a wave-local RMX_SYNC()"
#endif

void launch_step_w2_64(const rmx_model* m, const rmx_batch* b, int integ, const DevOpts& o, const StepArgs& a) {
    const dim3 grid(b->B), block(128);
    b->last_kernel = integ == INTEG_BDF1 ? "k_step_bdf1<64,w2>" : "k_step_bdf2<64,w2>";
    const size_t smem_bytes = m->smem_bytes + ((sizeof(double) * W2_HELP_DOUBLES + 15) & ~(size_t)15);      // + the helper wave's own area
    if (m->dm.is_chain && m->dm.n == RMX_NP) {      // a serial chain that fills every node slot: FULLCHAIN (no tree paths in the front)
        if (integ == INTEG_BDF1) RMX_LAUNCH((k_step_bdf1<RMX_NP, false, false, true, TAG_W2>), grid, block, smem_bytes, b->stream, m->dm, o, a);
        else RMX_LAUNCH((k_step_bdf2<RMX_NP, false, false, true, TAG_W2>), grid, block, smem_bytes, b->stream, m->dm, o, a);
        return;
    }
    if (m->dm.n == RMX_NP) {          // every node slot in use: the n == NP instantiation
        // (BDF1 without an energy record - the benchmark's launch -: the instantiation that does not carry the last evaluation's energies)
        if (integ == INTEG_BDF1 && !a.histT) RMX_LAUNCH((k_step_bdf1<RMX_NP, false, false, false, TAG_W2_NOE>), grid, block, smem_bytes, b->stream, m->dm, o, a);
        else if (integ == INTEG_BDF1) RMX_LAUNCH((k_step_bdf1<RMX_NP, false, false, false, TAG_W2>), grid, block, smem_bytes, b->stream, m->dm, o, a);
        else RMX_LAUNCH((k_step_bdf2<RMX_NP, false, false, false, TAG_W2>), grid, block, smem_bytes, b->stream, m->dm, o, a);
        return;
    }
    if (integ == INTEG_BDF1) RMX_LAUNCH((k_step_bdf1<RMX_NP, false, false, false, TAG_W2 + 1>), grid, block, smem_bytes, b->stream, m->dm, o, a);
    else RMX_LAUNCH((k_step_bdf2<RMX_NP, false, false, false, TAG_W2 + 1>), grid, block, smem_bytes, b->stream, m->dm, o, a);
}

#elif RMX_PART == 6      // full 32-link serial chains, BDF1, two wavefronts per rollout: the second one evaluates the point that may end a solve
#if RMX_NP != 32 || !RMX_W2
#error "RMX_PART 6 is compiled for RMX_NP = 32 with -DRMX_W2=1 and a wave-local RMX_SYNC()"
#endif

void launch_step_w2c_32(const rmx_model* m, const rmx_batch* b, const DevOpts& o, const StepArgs& a) {
    const dim3 grid(b->B), block(128);
    b->last_kernel = "k_step_bdf1<32,fullchain,w2>";
    const size_t smem_bytes = m->smem_bytes + ((sizeof(double) * W2C_HELP_DOUBLES + 15) & ~(size_t)15);      // + the helper wave's own area
    RMX_LAUNCH((k_step_bdf1<RMX_NP, false, false, true, TAG_W2>), grid, block, smem_bytes, b->stream, m->dm, o, a);
}

#elif RMX_PART == 7      // the full 32-link serial chain, BDF1: two points per evaluation of the front (rmx_pair32.h)
#if RMX_NP != 32
#error "RMX_PART 7 is compiled for RMX_NP = 32"
#endif
#include "rmx_pair32.h"

template <bool ENERGY>
__global__ void __launch_bounds__(64) k_step_bdf1_pair32(const DevModel Min, const DevOpts o, const StepArgs a) {
    constexpr int NP = 32;
    const DevModel M = model_view<NP, true>(Min);
    const unsigned long long tick0 = __builtin_amdgcn_s_memtime();
    const int traj = blockIdx.x;
    double *sAcc, *sCol;
    smem_setup<NP>(M, sAcc, sCol);
    const int lane = threadIdx.x;
    const int id = M.idx[lane & 31];             // both half-waves hold the chain: node = lane & 31
    const size_t off = (size_t)traj * M.nr + (id >= 0 ? id : 0);
    double q = id >= 0 ? a.q[off] : 0.0;
    double qd = id >= 0 ? a.qd[off] : 0.0;
    int iters = 0, halv = 0, status = 0;
    PivotPolicy piv;
    pair_rollout_bdf1<ENERGY>(M, o, a, sAcc, lane, traj, id, off, q, qd, iters, halv, status, piv);
    if (id >= 0 && lane < 32) {
        a.q[off] = q;
        a.qd[off] = qd;
    }
    if (lane == 0 && a.it) {
        a.it[traj] += iters;
        a.ls[traj] += halv;
        a.status[traj] |= status;
    }
    if (lane == 0 && a.ticks) a.ticks[traj] = __builtin_amdgcn_s_memtime() - tick0;      // (stored, not added: launch_step skips the fill for this kernel)
}

// Profiling hook (rmx_profile_phases) for the full 32-link chain: shader-clock cycles of the stages of one Newton iteration of the
// kernel above, measured in place with ITS device functions (the pair front with the LDS scan, the matrix-core Hessian stage staged
// from a half-wave, the guarded column-split solve) at the production occupancy.  out[16 traj + ..]: 0 one front, 1 front + Hessian
// stage, 2 solve, 3 the loop's own arithmetic (the two points, both norms, the compensated update); 4.. stamps inside the front and the
// Hessian stage (numbered as eval_front_e2 / eval_hess number them).
__global__ void __launch_bounds__(64) k_phase_time_pair32(const DevModel Min, const int reps, const double* __restrict__ q,
                                                          const double* __restrict__ qd, const double h, unsigned long long* __restrict__ out) {
    constexpr int NP = 32;
    const DevModel M = model_view<NP, true>(Min);
    double *sAcc, *sCol;
    smem_setup<NP>(M, sAcc, sCol);
    const int lane = threadIdx.x, traj = blockIdx.x;
    const int id = M.idx[lane & 31];
    const size_t off = (size_t)traj * M.nr + (id >= 0 ? id : 0);
    const double q0 = id >= 0 ? q[off] : 0.0, qd0 = id >= 0 ? qd[off] : 0.0;
    const double* cK = RMX_CONSTS(sAcc, M.n, NP);
    const double grav[3] = {M.grav[0], M.grav[1], M.grav[2]};
    double x = fma(h, qd0, q0), lo = 0.0;
    const double qB = x;
    unsigned long long tg = 0, tH = 0, tLU = 0, tred = 0;
    unsigned long long stamps[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    double sink = 0.0;
    int prim = 0;
    for (int r = 0; r < reps; ++r) {
        NodeOut e;
        FrontState fs;
        bool ta, tb;
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        const double qdn = ((x - q0) + lo) / h;
        const double xn = fma(h, qdn, x);
        const bool isQ = (lane >> 5) != prim;
        const double xe = isQ ? xn : x;
        const double xqd = isQ ? ((xn - x) + 0.0) / h : qdn;
        const double xv = isQ ? ((xn - xn) + 0.0) : ((x - qB) + lo);
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        eval_front_pair<false, true>(M.n, cK, grav, lane, xe, xqd, xv, h, e, fs, ta, tb, sAcc);
        sink += e.g;
        const unsigned long long t2 = __builtin_amdgcn_s_memtime();
        eval_front_pair<false, true, true>(M.n, cK, grav, lane, xe, xqd, xv, h, e, fs, ta, tb, sAcc, stamps);
        double Hdummy[NP];
        (void)eval_hess<NP, true, false, false, true>(M, lane, fs, Hdummy, stamps, sAcc, e.g, prim);
        const unsigned long long t3 = __builtin_amdgcn_s_memtime();
        bool lu_ok;
        double dx = lu_solve_neg_diag32(M.n, lane, sAcc, e.g, lu_ok);
        sink += lu_ok ? 0.0 : 1.0;
        const unsigned long long t4 = __builtin_amdgcn_s_memtime();
        dx = dup_lo(dx);
        double ga2, gb2;
        wave_sum_dual(e.g * e.g, ga2, gb2);
        const double dxn2 = wave_sum_np<NP>(dx * dx);
        sink += (prim ? gb2 : ga2) + dxn2;
        double xs, ls;
        two_sum(x, fma(1e-3, dx, lo), xs, ls);      // keep the iterations data dependent
        x = xs;
        lo = ls;
        prim ^= 1;
        const unsigned long long t5 = __builtin_amdgcn_s_memtime();
        tg += t2 - t1; tH += t3 - t2; tLU += t4 - t3; tred += (t1 - t0) + (t5 - t4);
    }
    if (lane == 0) {
        out[16 * traj + 0] = tg; out[16 * traj + 1] = tH; out[16 * traj + 2] = tLU; out[16 * traj + 3] = tred;
        for (int k = 0; k < 12; ++k) out[16 * traj + 4 + k] = stamps[k];
    }
    if (sink == 1.2345e301) out[0] = 0;   // keep the results live
}

void launch_phase_pairchain_32(const rmx_model* m, const rmx_batch* b, int reps, double h, unsigned long long* d) {
    const dim3 grid(b->B), block(64);
    RMX_LAUNCH(k_phase_time_pair32, grid, block, m->smem_bytes, b->stream, m->dm, reps, b->q, b->qd, h, d);
}

void launch_step_pairchain_32(const rmx_model* m, const rmx_batch* b, const DevOpts& o, const StepArgs& a) {
    const dim3 grid(b->B), block(64);
    b->last_kernel = "k_step_bdf1_pair32";
    // (a call that records T, V per step takes the instantiation that carries the energies of the last evaluation)
    if (a.histT) RMX_LAUNCH(k_step_bdf1_pair32<true>, grid, block, m->smem_bytes, b->stream, m->dm, o, a);
    else RMX_LAUNCH(k_step_bdf1_pair32<false>, grid, block, m->smem_bytes, b->stream, m->dm, o, a);
}

#elif RMX_PART == 8      // adjoint forward sweep of trees of <= 16 nodes with a second wavefront per rollout for M, D (k_adjoint_fwd HELP)
#if RMX_NP != 16
#error "RMX_PART 8 is compiled for RMX_NP = 16 with a wave-local RMX_SYNC()"
#endif

void launch_adjoint_help_16(const rmx_model* m, const rmx_batch* b, int integ, const DevOpts& o, const AdjArgs& a) {
    const dim3 grid(b->B);
    const size_t smem_bytes = m->smem_bytes + sizeof(double) * adj_hand_doubles(RMX_NP);
    if (integ == INTEG_BDF1 && m->dm.is_chain && m->dm.n == RMX_NP) {      // (configs[3]: the full 16-link chain)
        RMX_LAUNCH((k_adjoint_fwd<RMX_NP, 1, true, true>), grid, dim3(128), smem_bytes, b->stream, m->dm, o, a);
        k_adjoint_bwd<RMX_NP, 1, true><<<grid, dim3(64), 0, b->stream>>>(m->dm, o, a);
    } else if (m->dm.is_chain && m->dm.n == RMX_NP) {
        RMX_LAUNCH((k_adjoint_fwd<RMX_NP, 2, true, true>), grid, dim3(128), smem_bytes, b->stream, m->dm, o, a);
        k_adjoint_bwd<RMX_NP, 2, true><<<grid, dim3(64), 0, b->stream>>>(m->dm, o, a);
    } else if (integ == INTEG_BDF1) {
        RMX_LAUNCH((k_adjoint_fwd<RMX_NP, 1, true>), grid, dim3(128), smem_bytes, b->stream, m->dm, o, a);
        k_adjoint_bwd<RMX_NP, 1><<<grid, dim3(64), 0, b->stream>>>(m->dm, o, a);
    } else {
        RMX_LAUNCH((k_adjoint_fwd<RMX_NP, 2, true>), grid, dim3(128), smem_bytes, b->stream, m->dm, o, a);
        k_adjoint_bwd<RMX_NP, 2><<<grid, dim3(64), 0, b->stream>>>(m->dm, o, a);
    }
}

#elif RMX_PART == 2      // the FULLCHAIN instantiations of the plain step kernels (sizes 16, 32, 64), one object per size

void RMX_CAT(launch_step_fullchain_, RMX_NP)(const rmx_model* m, const rmx_batch* b, int integ, const DevOpts& o, const StepArgs& a) {
    const dim3 grid(b->B), block(64);
    b->last_kernel = integ == INTEG_BDF1 ? "k_step_bdf1<" RMX_STR(RMX_NP) ",fullchain>" : "k_step_bdf2<" RMX_STR(RMX_NP) ",fullchain>";
    if (integ == INTEG_BDF1) RMX_LAUNCH((k_step_bdf1<RMX_NP, false, false, true>), grid, block, m->smem_bytes, b->stream, m->dm, o, a);
    else RMX_LAUNCH((k_step_bdf2<RMX_NP, false, false, true>), grid, block, m->smem_bytes, b->stream, m->dm, o, a);
}
#if RMX_NP == 64
// a tree that fills all 64 node slots (n == NP at compile time; LDS-resident constants)
void launch_step_fulln_64(const rmx_model* m, const rmx_batch* b, int integ, const DevOpts& o, const StepArgs& a) {
    const dim3 grid(b->B), block(64);
    b->last_kernel = integ == INTEG_BDF1 ? "k_step_bdf1<64,fulln>" : "k_step_bdf2<64,fulln>";
    if (integ == INTEG_BDF1) RMX_LAUNCH((k_step_bdf1<RMX_NP, false, false, false, TAG_FULLN>), grid, block, m->smem_bytes, b->stream, m->dm, o, a);
    else RMX_LAUNCH((k_step_bdf2<RMX_NP, false, false, false, TAG_FULLN>), grid, block, m->smem_bytes, b->stream, m->dm, o, a);
}
#endif

#elif RMX_PART == 1

void RMX_CAT(launch_eval_ct_, RMX_NP)(const rmx_model* m, const rmx_batch* b, bool wantH, double eta, double* dg, double* dH) {
    const dim3 grid(b->B), block(64);
    if (wantH) RMX_LAUNCH((k_eval<RMX_NP, true, true>), grid, block, m->smem_bytes, b->stream, m->dm, b->B, b->tmpA, b->tmpB, b->tmpC, eta, dg, dH, b->chart);
    else RMX_LAUNCH((k_eval<RMX_NP, false, true>), grid, block, m->smem_bytes, b->stream, m->dm, b->B, b->tmpA, b->tmpB, b->tmpC, eta, dg, dH, b->chart);
}
void RMX_CAT(launch_mfd_ct_, RMX_NP)(const rmx_model* m, const rmx_batch* b, double* dM, double* df, double* dD) {
    const dim3 grid(b->B), block(64);
    RMX_LAUNCH((k_eval_mfd<RMX_NP, true>), grid, block, m->smem_bytes, b->stream, m->dm, b->B, b->tmpA, b->tmpB, dM, df, dD, b->chart);
}
void RMX_CAT(launch_step_ct_, RMX_NP)(const rmx_model* m, const rmx_batch* b, int integ, const DevOpts& o, const StepArgs& a) {
    const dim3 grid(b->B), block(64);
    b->last_kernel = integ == INTEG_BDF1 ? "k_step_bdf1<" RMX_STR(RMX_NP) ",ct>" : "k_step_bdf2<" RMX_STR(RMX_NP) ",ct>";      // (lean launch + launch with the contact terms)
#if RMX_NP == 32
    // serial chains with ForceGroundCuboid (no Euler-chart joints): free flight, contact and the cooperative groups in ONE launch
    if (m->pair32 && m->dm.con && a.fused) return launch_step_pair_32(m, b, integ, o, a, true);
#endif
    // every trajectory as far as it stays clear of the ground (all the way in scenes without ForceGroundCuboid) ...
    if (integ == INTEG_BDF1) RMX_LAUNCH((k_step_bdf1<RMX_NP, true, true>), grid, block, m->smem_bytes, b->stream, m->dm, o, a);
    else RMX_LAUNCH((k_step_bdf2<RMX_NP, true, true>), grid, block, m->smem_bytes, b->stream, m->dm, o, a);
    if (!m->dm.con) return;
#if RMX_NP == 32
    // serial chains (no Euler-chart joints): the kernels around newton_pair (RMX_PART 4, rmx_ct32.h) take the rest of the steps with the
    // contact terms, and what they park (a Newton solve that keeps running out its line searches) in cooperative groups
    if (m->pair32) return launch_step_pair_32(m, b, integ, o, a, false);
#endif
    // ... and the rest of its steps with the contact terms
    if (integ == INTEG_BDF1) RMX_LAUNCH((k_step_bdf1<RMX_NP, true>), grid, block, m->smem_bytes, b->stream, m->dm, o, a);
    else RMX_LAUNCH((k_step_bdf2<RMX_NP, true>), grid, block, m->smem_bytes, b->stream, m->dm, o, a);
}
void RMX_CAT(launch_energy_ct_, RMX_NP)(const rmx_model* m, const rmx_batch* b, double* dT, double* dV) {
    const dim3 grid(b->B), block(64);
    RMX_LAUNCH((k_energy<RMX_NP, true>), grid, block, m->smem_bytes, b->stream, m->dm, b->B, b->q, b->qd, dT, dV, b->chart);
}

#else

void RMX_CAT(launch_eval_, RMX_NP)(const rmx_model* m, const rmx_batch* b, bool wantH, double eta, double* dg, double* dH) {
    const dim3 grid(b->B), block(64);
    // scenes with ForceGroundCuboid or JointSpherical run the extended instantiations (CT), everything else the plain ones
    if (m->dm.con != nullptr || m->dm.nsph > 0) return RMX_CAT(launch_eval_ct_, RMX_NP)(m, b, wantH, eta, dg, dH);
    if (wantH) RMX_LAUNCH((k_eval<RMX_NP, true, false>), grid, block, m->smem_bytes, b->stream, m->dm, b->B, b->tmpA, b->tmpB, b->tmpC, eta, dg, dH, nullptr);
    else RMX_LAUNCH((k_eval<RMX_NP, false, false>), grid, block, m->smem_bytes, b->stream, m->dm, b->B, b->tmpA, b->tmpB, b->tmpC, eta, dg, dH, nullptr);
}

void RMX_CAT(launch_step_np_, RMX_NP)(const rmx_model* m, const rmx_batch* b, int integ, const DevOpts& o, const StepArgs& a) {
    const dim3 grid(b->B), block(64);
    b->last_kernel = integ == INTEG_BDF1 ? "k_step_bdf1<" RMX_STR(RMX_NP) ">" : "k_step_bdf2<" RMX_STR(RMX_NP) ">";      // (the launchers taken below overwrite it)
    if (m->dm.con != nullptr || m->dm.nsph > 0) return RMX_CAT(launch_step_ct_, RMX_NP)(m, b, integ, o, a);
#if RMX_NP == 64
    // 33..64 nodes in a batch of at most one rollout per two SIMDs: a second wavefront per rollout for the Hessian and the solve
    if (m->w2_max_batch > 0 && b->B <= m->w2_max_batch) return launch_step_w2_64(m, b, integ, o, a);
#endif
#if RMX_NP == 32
    // the full 32-link chain under BDF1 (BASELINE.json configs[1]): two points per evaluation, the second one the next step's first
    // (rmx_pair32.h; bit-identical to the one-point kernels below, which RMX_PAIRC=0 keeps reachable)
    if (m->dm.is_chain && m->dm.n == RMX_NP && integ == INTEG_BDF1 && a.pairc) return launch_step_pairchain_32(m, b, o, a);
    // the full 32-link chain in a shard of one rollout per two SIMDs or fewer (the 1024-rollout batch on two or more GPUs): a second
    // wavefront per rollout evaluates the point that may end a step's solve while the first evaluates the next step's first point
    if (m->dm.is_chain && m->dm.n == RMX_NP && integ == INTEG_BDF1 && m->w2_max_batch > 0 && b->B >= m->w2_min_batch && b->B <= m->w2_max_batch)
        return launch_step_w2c_32(m, b, o, a);
#endif
#if RMX_NP >= 16 && !defined(RMX_NO_FULLCHAIN)      // (the macro: development aid, tools/build_variant.py)
    if (m->dm.is_chain && m->dm.n == RMX_NP) return RMX_CAT(launch_step_fullchain_, RMX_NP)(m, b, integ, o, a);
#endif
#if RMX_NP == 64
    // More than two rollouts per CU: the kernels that read the per-node constants from global memory (33.8 KB of LDS per wavefront
    // instead of 68.6 KB: four wavefronts per CU instead of two).  Up to two per CU the LDS-resident constants are faster (-7 %).
    if (m->dm.gconst && m->gconst_min_batch > 0 && b->B >= m->gconst_min_batch) return launch_step_gconst_64(m, b, integ, o, a);
#if !defined(RMX_NO_FULLCHAIN)
    if (m->dm.n == RMX_NP) return launch_step_fulln_64(m, b, integ, o, a);
#endif
#endif
    if (integ == INTEG_BDF1) RMX_LAUNCH((k_step_bdf1<RMX_NP, false>), grid, block, m->smem_bytes, b->stream, m->dm, o, a);
    else RMX_LAUNCH((k_step_bdf2<RMX_NP, false>), grid, block, m->smem_bytes, b->stream, m->dm, o, a);
}

void RMX_CAT(launch_euler_, RMX_NP)(const rmx_model* m, const rmx_batch* b, double h, const StepArgs& a) {
    const dim3 grid(b->B), block(64);
    RMX_LAUNCH((k_step_euler<RMX_NP>), grid, block, m->smem_bytes, b->stream, m->dm, h, a);
}

void RMX_CAT(launch_energy_, RMX_NP)(const rmx_model* m, const rmx_batch* b, double* dT, double* dV) {
    const dim3 grid(b->B), block(64);
    if (m->dm.con || m->dm.nsph) return RMX_CAT(launch_energy_ct_, RMX_NP)(m, b, dT, dV);
    RMX_LAUNCH((k_energy<RMX_NP, false>), grid, block, m->smem_bytes, b->stream, m->dm, b->B, b->q, b->qd, dT, dV, nullptr);
}

void RMX_CAT(launch_adjoint_, RMX_NP)(const rmx_model* m, const rmx_batch* b, int integ, const DevOpts& o, const AdjArgs& a) {
    const dim3 grid(b->B), block(64);
#if RMX_NP == 16
    // up to one rollout per two SIMDs: a second wavefront per rollout forms and stores M, D (RMX_PART 8; RMX_ADJ_HELP=0: tests)
    {
        const char* ah = getenv("RMX_ADJ_HELP");      // (read at every call: tests switch it inside one process)
        if (!(ah && atoi(ah) == 0) && m->adj_help_max_batch > 0 && b->B <= m->adj_help_max_batch) return launch_adjoint_help_16(m, b, integ, o, a);
    }
#endif
#if RMX_NP == 16
    if (integ == INTEG_BDF1 && m->dm.is_chain && m->dm.n == RMX_NP) {      // the full 16-link chain: the instantiation RMX_PART 8 runs with its helper wave
        RMX_LAUNCH((k_adjoint_fwd<RMX_NP, 1, false, true>), grid, block, m->smem_bytes, b->stream, m->dm, o, a);
        k_adjoint_bwd<RMX_NP, 1, true><<<grid, block, 0, b->stream>>>(m->dm, o, a);
        return;
    }
    if (m->dm.is_chain && m->dm.n == RMX_NP) {
        RMX_LAUNCH((k_adjoint_fwd<RMX_NP, 2, false, true>), grid, block, m->smem_bytes, b->stream, m->dm, o, a);
        k_adjoint_bwd<RMX_NP, 2, true><<<grid, block, 0, b->stream>>>(m->dm, o, a);
        return;
    }
#endif
    if (integ == INTEG_BDF1) {
        RMX_LAUNCH((k_adjoint_fwd<RMX_NP, 1>), grid, block, m->smem_bytes, b->stream, m->dm, o, a);
        k_adjoint_bwd<RMX_NP, 1><<<grid, block, 0, b->stream>>>(m->dm, o, a);
    } else {
        RMX_LAUNCH((k_adjoint_fwd<RMX_NP, 2>), grid, block, m->smem_bytes, b->stream, m->dm, o, a);
        k_adjoint_bwd<RMX_NP, 2><<<grid, block, 0, b->stream>>>(m->dm, o, a);
    }
}

void RMX_CAT(launch_mfd_, RMX_NP)(const rmx_model* m, const rmx_batch* b, double* dM, double* df, double* dD) {
    if (m->dm.con) return RMX_CAT(launch_mfd_ct_, RMX_NP)(m, b, dM, df, dD);
    const dim3 grid(b->B), block(64);
    RMX_LAUNCH((k_eval_mfd<RMX_NP>), grid, block, m->smem_bytes, b->stream, m->dm, b->B, b->tmpA, b->tmpB, dM, df, dD, b->chart);
}

#if RMX_NP == 64
void launch_stage_consts_64(const rmx_model* m, double* dst, hipStream_t stream) {
    k_stage_consts<64><<<dim3(1), dim3(64), 0, stream>>>(m->dm, dst);
}
#endif

void RMX_CAT(launch_phase_, RMX_NP)(const rmx_model* m, const rmx_batch* b, int reps, double h, unsigned long long* d) {
    const dim3 grid(b->B), block(64);
#if RMX_NP == 32
    // the full 32-link chain: the stages of the kernel that runs it (RMX_PART 7); everything else: the generic device functions below
    if (m->dm.is_chain && m->dm.n == RMX_NP && !m->dm.con && !m->dm.nsph) return launch_phase_pairchain_32(m, b, reps, h, d);
#endif
    RMX_LAUNCH((k_phase_time<RMX_NP>), grid, block, m->smem_bytes, b->stream, m->dm, reps, b->q, b->qd, h, d);
}

#endif
