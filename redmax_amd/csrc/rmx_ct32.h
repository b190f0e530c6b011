// rmx_ct32.h -- serial chains of at most 32 nodes with ForceGroundCuboid (BASELINE.json configs[4]): the Newton solve of the step
// kernels with ONE evaluation site, laid out for the two things that decide this workload's launch time.
//
// (1) A tree of <= 32 nodes leaves lanes 32..63 idle in every lane = node stage.  eval_front_pair is the FULL front (residual,
//     energies and the 28 subtree sums + per-node state the Hessian stage consumes) for TWO iterates at once: lanes 0..31 carry
//     trial point a, lanes 32..63 trial point b, node = lane & 31 in both halves, every stage lane-local or inside 16-lane DPP
//     rows (the subtree sums as register scans: no LDS, no barrier).  The line search of newton()
//     (driverRedMaxBDF2.m, the same as driverRedMaxBDF1.m:123-141) tests alpha = 1, 1/2, 1/4 ... IN ORDER, but its trial residuals
//     are independent: every evaluation of newton_pair takes the next two trials, the decisions are walked in the reference's
//     order, and the trial that ends the search already holds the state the Hessian stage needs (moved down from lanes 32..63
//     when it was b): no re-evaluation of the accepted point, half the evaluations of a search that runs out its 20 trials.
// (2) newton_impl / newton_rot of rmx_device.h instantiate the front two to five times per solve, twice over (pivot-only and
//     guarded), three solves per BDF2 step: 146 600 instructions, 1 074 spilled SGPRs and 836 bytes of scratch per lane in
//     k_step_bdf2<32, true>.  Here the loop is rotated around a single front, a single Hessian stage and the two solves (the
//     pivoting one behind a wave-uniform flag); the kernel (rmx_ct32.hip) calls it from one site for every solve of every
//     integrator.
//
// COOP: the cooperative launch (park and relaunch, see CoopCtx in rmx_device.h) in its second form.  A group of COOP_G wavefronts
// takes ALL 2 COOP_G trials of a line search in one evaluation (member m: trials 1 + 2 m and 2 + 2 m), exchanges one decision
// word per member, and the member that holds the accepted trial - the WINNER - alone runs the Hessian stage and the solve and
// publishes dx and |g|^2 (COOP_REC doubles through global memory, in self-tagged words: no fence); the others recompute the
// accepted iterate from x0, alpha and the old dx, which they all hold bit for bit.  Evaluations nobody distributes (the first one
// of a solve, the re-evaluation before a pivoting solve) are run by every member redundantly: same code, same inputs, same bits.
#pragma once

namespace rmx {

// measurement builds (tools/build_variant.py --part 4 -- -DRMX_TICK_PHASE=k): RMX_TICK_PHASE=k makes rmx_step_ticks report the shader-clock ticks of ONE phase of
// newton_pair instead of the rollout's whole share of the launch: 1 front, 2 Hessian stage, 3 solve, 4 exchange, 5 wait for the winner,
// 6 publish
#ifdef RMX_TICK_PHASE
#define RMX_PH_BEGIN(k) const unsigned long long ph_t##k = (RMX_TICK_PHASE == k) ? __builtin_amdgcn_s_memtime() : 0ull;
#define RMX_PH_END(k) if (RMX_TICK_PHASE == k) cx.phase += __builtin_amdgcn_s_memtime() - ph_t##k;
#else
#define RMX_PH_BEGIN(k)
#define RMX_PH_END(k)
#endif

// suffix sums of NS numbers per node along two chains of 32 nodes side by side (lanes 0..31, 32..63): row_shl scans inside the
// 16-lane rows, then rows 0 and 2 add the complete total of lane 16 / 48
template <int NS, int NA>
__device__ __forceinline__ void chain_suffix_sum_pair(const int lane, double (&S)[NA]) {
    static_assert(NS <= NA, "scan wider than the array");
#pragma unroll
    for (int c = 0; c < NS; ++c) S[c] += dpp_shl0<1>(S[c]);
#pragma unroll
    for (int c = 0; c < NS; ++c) S[c] += dpp_shl0<2>(S[c]);
#pragma unroll
    for (int c = 0; c < NS; ++c) S[c] += dpp_shl0<4>(S[c]);
#pragma unroll
    for (int c = 0; c < NS; ++c) S[c] += dpp_shl0<8>(S[c]);
    const double w0 = ((lane >> 4) == 0) ? 1.0 : 0.0, w2 = ((lane >> 4) == 2) ? 1.0 : 0.0;
    // broadcasts ahead of their FMAs (readlane -> use hazard), in batches the SGPR file holds
    constexpr int NB = NS % 7 == 0 ? 7 : 6;
    static_assert(NS % NB == 0, "batching");
#pragma unroll
    for (int c0 = 0; c0 < NS; c0 += NB) {
        double t0[NB], t2[NB];
#pragma unroll
        for (int c = 0; c < NB; ++c) {
            t0[c] = readlane_d(S[c0 + c], 16);
            t2[c] = readlane_d(S[c0 + c], 48);
        }
#pragma unroll
        for (int c = 0; c < NB; ++c) S[c0 + c] = fma(w2, t2[c], fma(w0, t0[c], S[c0 + c]));
    }
}

// The same sums for a FULL chain of 32 nodes through the accumulation scratch, in the order of the one-point front
// (eval_front_e2<32, true>: per component the serial suffix over nodes 31..16, the serial suffix over nodes 15..0, and the upper
// half's total added to every lower node): each half's sums carry the bits that front produces for the same point.  Trial point
// t = lane >> 5 owns rows [32 t, 32 t + 32) of the scratch (2 x 32 x ACC_STRIDE doubles <= acc_doubles(32, 32)); lane = node writes
// its 28 numbers, lane = component (of its own trial point) scans the 32 nodes in registers, lane = node reads the sums back.
template <int NS>
__device__ __forceinline__ void chain_suffix_sum_pair_lds(double* __restrict__ sAcc, const int lane, double (&S)[NACC]) {
    static_assert(2 * 32 * ACC_STRIDE <= acc_doubles(32, 32), "the pair scan needs two blocks of 32 accumulation rows");
    constexpr int AS = ACC_STRIDE;
    const int jc = lane & 31;
    double* const blk = sAcc + (lane >> 5) * (32 * AS);
    {
        double* A = blk + jc * AS;
#pragma unroll
        for (int c = 0; c < NS; ++c) A[c] = S[c];
    }
    RMX_SYNC();
    {
        double a[32];
#pragma unroll
        for (int t = 0; t < 32; ++t) a[t] = blk[t * AS + jc];      // components >= NS: finite junk, never stored
        // every read in flight before the first addition: left alone, the scheduler issues the reads two ahead of the serial chains
        // that consume them, and the lone wavefront pays one LDS round trip per pair of nodes
#ifndef RMX_PAIR_SCAN_NOSB
        __builtin_amdgcn_sched_barrier(0);
#endif
        double up = 0.0, dn = 0.0;
#pragma unroll
        for (int t = 15; t >= 0; --t) {                            // (two independent serial chains)
            up += a[16 + t];
            a[16 + t] = up;
            dn += a[t];
            a[t] = dn;
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) a[t] += up;
        if (jc < NS) {
#pragma unroll
            for (int t = 0; t < 32; ++t) blk[t * AS + jc] = a[t];
        }
    }
    RMX_SYNC();
    {
        const double* A = blk + jc * AS;
#pragma unroll
        for (int c = 0; c < NS; ++c) S[c] = A[c];
    }
    RMX_SYNC();
}

// The full front of two iterates of a serial chain of n <= 32 nodes.  xq, xqd, xv: the coordinates of node lane & 31 at trial point
// lane >> 5 (zeros where the node has no DOF).  Same formulas, in the same order, as eval_front_e2<32, true, false, CT>; the subtree
// sums by the register scan.  out / fs: per lane, i.e. for the lane's own trial point; fs.act is lane < n (what the Hessian stage,
// which works on lanes 0..31, expects), fs.touched is left to the caller (ta / tb: some corner of the tree penetrates at a / b).
// LDS_SCAN (the plain full chain, rmx_pair32.h): the 28 subtree sums through the accumulation scratch in the summation order of
// eval_front_e2<32, true> (chain_suffix_sum_pair_lds), so that each half reproduces the one-point front bit for bit.
// TIMED (k_phase_time_pair32): s_memtime stamps per stage, numbered as eval_front_e2's.
// REGK (rmx_pair32.h): the lane's 57 per-node constants (PAIR_NK rows: K 36, sb 6, I4 4, prm 8, type, rel 2 - the first rows of the
// LDS table, in its order) come from the caller's registers (rk, loaded once per rollout by pair_load_consts) instead of the wave's
// LDS copy: 54 ds_read_b64 and 27 KB of LDS traffic less per evaluation.
constexpr int PAIR_NK = 57;
template <bool CT, bool LDS_SCAN = false, bool TIMED = false, bool REGK = false>
__device__ __forceinline__ void eval_front_pair(const int n, const double* __restrict__ cK, const double (&grav)[3], const int lane,
                                                const double xq, const double xqd, const double xv, const double eta,
                                                NodeOut& out, FrontState& fs, bool& ta, bool& tb, double* __restrict__ sAcc = nullptr,
                                                unsigned long long* stamps = nullptr, const double* rk = nullptr) {
    static_assert(!(REGK && CT), "REGK: the plain constants only");
#define RMX_PK(row) (REGK ? rk[(row)] : cK[(row) * CS + jc])
    unsigned long long last_ = TIMED ? __builtin_amdgcn_s_memtime() : 0ull;
    constexpr int NP = 32;
    constexpr int CS = cstride(NP);
    const double e2 = eta * eta;
    const int jc = lane & 31;
    const double* cCon = cK + (36 + 6 + 4 + 8 + 1 + 2 + MAXROUNDS + 1) * CS;
    const int type = (int)RMX_PK(54);
    const bool dof = type != 0;
    fs.anc_m = (unsigned long long)__double_as_longlong(RMX_PK(55));
    fs.desc_m = (unsigned long long)__double_as_longlong(RMX_PK(56));
    const double q = xq, qd = xqd, v = xv;
    double u = 0.0, w = 0.0;
    if (type == 1) {
        sincos(q, &u, &w);
    } else if (type == 2) {
        u = q;
    }
    double R[9], p[3];
#pragma unroll
    for (int c = 0; c < 9; ++c) R[c] = RMX_PK(c) + u * RMX_PK(12 + c) + w * RMX_PK(24 + c);
#pragma unroll
    for (int c = 0; c < 3; ++c) p[c] = RMX_PK(9 + c) + u * RMX_PK(21 + c) + w * RMX_PK(33 + c);
    RMX_STAMP(0)
    chain_scan_transform_dual(lane, R, p);
    RMX_STAMP(1)
    double sbw[3], sbv[3], t3[3];
    double (&sw)[3] = fs.sw;
    double (&sv)[3] = fs.sv;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        sbw[c] = RMX_PK(36 + c);
        sbv[c] = RMX_PK(39 + c);
    }
    mat3v(R, sbw, sw);
    mat3v(R, sbv, sv);
    cross3(p, sw, t3);
#pragma unroll
    for (int c = 0; c < 3; ++c) sv[c] += t3[c];
    double (&phw)[3] = fs.phw;
    double (&phv)[3] = fs.phv;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        phw[c] = sw[c] * qd;
        phv[c] = sv[c] * qd;
    }
    chain_scan_sum6_dual(lane, phw, phv);
    RMX_STAMP(2)
    double (&xiw)[3] = fs.xiw;
    double (&xiv)[3] = fs.xiv;
    double (&bw)[3] = fs.bw;
    double (&bv)[3] = fs.bv;
    cross3(phw, sw, xiw);
    cross3(phv, sw, xiv);
    cross3(phw, sv, t3);
#pragma unroll
    for (int c = 0; c < 3; ++c) xiv[c] += t3[c];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        bw[c] = sw[c] * v + e2 * qd * xiw[c];
        bv[c] = sv[c] * v + e2 * qd * xiv[c];
    }
    chain_scan_sum6_dual(lane, bw, bv);
    RMX_STAMP(3)
    const double I1 = RMX_PK(42), I2 = RMX_PK(43);
    const double I3 = RMX_PK(44), ms = RMX_PK(45);
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
    double ht[3], hf[3], bt[3], bf[3];
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
    double fct[3], fcf[3], a3[3], b3[3];
    cross3(phw, ht, a3);
    cross3(phv, hf, b3);
    cross3(phw, hf, fcf);
    const double gv[3] = {grav[0], grav[1], grav[2]};
    double fgt[3];
    cross3(mc, gv, fgt);
    double wt[3], wf[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        fct[c] = -a3[c] - b3[c];
        wt[c] = bt[c] - e2 * (fct[c] + fgt[c]);
        wf[c] = bf[c] - e2 * (-fcf[c] + ms * gv[c]);
    }
    double eVc = 0.0;
    ta = tb = false;
    if constexpr (CT) {
        const bool con = cCon[jc] != 0.0;
        const double sd[3] = {cCon[CS + jc], cCon[2 * CS + jc], cCon[3 * CS + jc]};
        GroundC G;
        {
            const double* gr = cK + NCONST * CS + jc;      // ground_of
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                G.n[c] = gr[c * CS];
                G.gx[c] = gr[(3 + c) * CS];
            }
            G.kn = gr[6 * CS];
            G.kt = gr[7 * CS];
            G.mu = gr[8 * CS];
            G.kdc = gr[9 * CS];
        }
        const double dc = G.n[0] * (p[0] - G.gx[0]) + G.n[1] * (p[1] - G.gx[1]) + G.n[2] * (p[2] - G.gx[2]);
        const double reach = 0.5 * (fabs(G.n[0] * R[0] + G.n[1] * R[3] + G.n[2] * R[6]) * sd[0] +
                                    fabs(G.n[0] * R[1] + G.n[1] * R[4] + G.n[2] * R[7]) * sd[1] +
                                    fabs(G.n[0] * R[2] + G.n[1] * R[5] + G.n[2] * R[8]) * sd[2]);
        const bool near = __any(con && !(dc - reach > 1e-9 * (fabs(dc) + reach)));
        if (near) {
            double Fc[6], k1[36];
            bool pen = false;
            contact_body<0>(G, con, sd, R, p, phw, phv, Fc, k1, eVc, &pen);
            const unsigned long long pm = __ballot(pen);
            ta = (unsigned)pm != 0u;
            tb = (unsigned)(pm >> 32) != 0u;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                wt[c] -= e2 * Fc[c];
                wf[c] -= e2 * Fc[3 + c];
            }
        }
    }
    const double stiff = RMX_PK(47), damp = RMX_PK(48);
    const double tau = RMX_PK(46), qRest = RMX_PK(49);
    const double qLimL = RMX_PK(50), qLimU = RMX_PK(51);
    const double qLimK = RMX_PK(52), qLimD = RMX_PK(53);
    const double hitL = (dof && q < qLimL) ? 1.0 : 0.0, hitU = (dof && q > qLimU) ? 1.0 : 0.0;
    {
        double eT = 0.5 * (dot3(phw, ht) + dot3(phv, hf));
        double eV = -dot3(gv, mc);
        if (dof) {
            const double dq = q - qRest;
            const double dqL = hitL * (qLimL - q), dqU = hitU * (qLimU - q);
            eV += 0.5 * stiff * (dq * dq) + 0.5 * qLimK * (dqL * dqL + dqU * dqU);
        }
        out.eT = eT;
        out.eV = eV + eVc;
    }
    double (&S)[NACC] = fs.S;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        S[c] = wt[c];
        S[3 + c] = wf[c];
    }
    S[6] = ms;
#pragma unroll
    for (int c = 0; c < 3; ++c) S[7 + c] = mc[c];
#pragma unroll
    for (int c = 0; c < 6; ++c) S[10 + c] = Ib[c];
    {
        // TL = X + X' + [h_tau],  X = Ibar [phi_w] + [mc][phi_v]   (see eval_front_e2)
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
                double t = 0.0;
#pragma unroll
                for (int l = 0; l < 3; ++l) t += Ibf[3 * i + l] * Om[3 * l + k] + Mc[3 * i + l] * Vx[3 * l + k];
                X[3 * i + k] = t;
            }
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int k = 0; k < 3; ++k) S[16 + 3 * i + k] = X[3 * i + k] + X[3 * k + i] + Ht[3 * i + k];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) S[25 + c] = hf[c];
    RMX_STAMP(4)
    if constexpr (LDS_SCAN) chain_suffix_sum_pair_lds<NACC>(sAcc, lane, S);
    else chain_suffix_sum_pair<NACC, NACC>(lane, S);
    RMX_STAMP(6)
    const double fr = (tau + fs.tau_add) + stiff * (qRest - q) - damp * qd + hitL * (qLimK * (qLimL - q) - qLimD * qd) +
                      hitU * (qLimK * (qLimU - q) - qLimD * qd);
    out.g = dof ? (dot3(sw, &S[0]) + dot3(sv, &S[3]) - e2 * fr) : 0.0;
#pragma unroll
    for (int c = 0; c < 9; ++c) fs.Rw[c] = R[c];
#pragma unroll
    for (int c = 0; c < 3; ++c) fs.pw[c] = p[c];
    fs.eta = eta;
    fs.kd = stiff + (hitL + hitU) * qLimK;
    fs.dd = damp + (hitL + hitU) * qLimD;
    fs.act = lane < n;
    fs.dof = dof;
    RMX_STAMP(8)
#undef RMX_PK
}

// the lane's constants for eval_front_pair<..., REGK = true>: rows 0 .. PAIR_NK - 1 of the wave's LDS table, column lane & 31
__device__ __forceinline__ void pair_load_consts(const double* __restrict__ cK, const int lane, double (&rk)[PAIR_NK]) {
    constexpr int CS = cstride(32);
    const int jc = lane & 31;
#pragma unroll
    for (int r = 0; r < 36 + 6 + 4 + 8 + 1; ++r) rk[r] = cK[r * CS + jc];
    rk[55] = cK[55 * CS + jc];
    rk[56] = cK[56 * CS + jc];
}


// the state of trial point b (lanes 32..63) into lanes 0..31, where the Hessian stage works
__device__ __forceinline__ void front_take_hi(FrontState& fs, NodeOut& e) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        fs.sw[c] = take_hi(fs.sw[c]); fs.sv[c] = take_hi(fs.sv[c]);
        fs.phw[c] = take_hi(fs.phw[c]); fs.phv[c] = take_hi(fs.phv[c]);
        fs.xiw[c] = take_hi(fs.xiw[c]); fs.xiv[c] = take_hi(fs.xiv[c]);
        fs.bw[c] = take_hi(fs.bw[c]); fs.bv[c] = take_hi(fs.bv[c]);
        fs.pw[c] = take_hi(fs.pw[c]);
    }
#pragma unroll
    for (int c = 0; c < NACC; ++c) fs.S[c] = take_hi(fs.S[c]);
#pragma unroll
    for (int c = 0; c < 9; ++c) fs.Rw[c] = take_hi(fs.Rw[c]);
    fs.kd = take_hi(fs.kd);
    fs.dd = take_hi(fs.dd);
    e.g = take_hi(e.g);
    e.eT = take_hi(e.eT);
    e.eV = take_hi(e.eV);
}

// H of the staging area (eval_hess<32, ..., ZERO_IDLE = false> has just left it there, row-major [32][HM_H_STRIDE]) as one row per
// lane for the pivoting solve, lanes >= 32 all-zero rows (its pivot search looks at every lane); hands sAcc back to the front
__device__ __forceinline__ void hess_rows_from_staging(const int n, const int lane, double* __restrict__ sAcc, double (&Hrow)[32]) {
    typedef double v2d __attribute__((ext_vector_type(2)));
    const v2d* hr = reinterpret_cast<const v2d*>(sAcc + (lane & 31) * HM_H_STRIDE);
    const bool lo_half = lane < 32;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const v2d t = hr[c];
        Hrow[2 * c] = lo_half ? t[0] : 0.0;
        Hrow[2 * c + 1] = lo_half ? t[1] : 0.0;
    }
    RMX_SYNC();
    if (lane < ACC_STRIDE) sAcc[n * ACC_STRIDE + lane] = 0.0;
    RMX_SYNC();
}

// ---- the cooperative group's second channel: what the winner of a line search publishes (see the header of this file)
constexpr int COOP_CODE_DX = 1, COOP_CODE_END = 2, COOP_CODE_REDO = 3, COOP_CODE_STALL = 4, COOP_CODE_WIDE = 5;      // (+ 16 x accepted trial + 4096 x status bits)
// No fence anywhere: an agent-scope release / acquire on this part writes back and invalidates an L2 (one per XCD) that the scratch
// traffic of every wavefront of the XCD lives in - measured, a fenced hand-over cost 60 us per line search.  Instead every 64-bit word of
// the record is complete in itself, like the decision words: the round it belongs to in its upper half, 32 bits of payload in its lower
// half, written and read with relaxed agent-scope atomics (single-copy atomic); a reader polls its own two words until both carry
// the current round.  Lane l < 32: dx of node l, lanes 32 .. 35: |g|^2, code, T, V.
struct CoopPub {
    unsigned long long* rec = nullptr;           // this group's 2 COOP_REC words
};
__device__ __forceinline__ void coop_publish(const CoopPub& pb, const unsigned round, const int lane, const double gn2, const int code,
                                             const double T, const double V, const double dx) {
    const double v = lane < 32 ? dx : (lane == 32 ? gn2 : (lane == 33 ? (double)code : (lane == 34 ? T : V)));
    if (lane < 36) {
        const unsigned long long tg = (unsigned long long)round << 32;
        __hip_atomic_store(pb.rec + 2 * lane, tg | (unsigned)__double2loint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(pb.rec + 2 * lane + 1, tg | (unsigned)__double2hiint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// false: the wait ran out (or the group's abort flag is up)
__device__ __forceinline__ bool coop_collect(const CoopPub& pb, const CoopCtx& cx, const unsigned round, const int lane, double& gn2, int& code,
                                             double& T, double& V, double& dx) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long w0 = 0ull, w1 = 0ull;
    while (true) {
        bool mine = true;
        if (lane < 36) {
            w0 = __hip_atomic_load(pb.rec + 2 * lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            w1 = __hip_atomic_load(pb.rec + 2 * lane + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            mine = (unsigned)(w0 >> 32) == round && (unsigned)(w1 >> 32) == round;
        }
        if (__all(mine)) break;
        unsigned ab = 0u;
        if (lane == 0) ab = coop_load(cx.words + 2 * COOP_G);
        if (__any(ab != 0u) || __builtin_amdgcn_s_memtime() - t0 > cx.ticks) {
            if (lane == 0) coop_store(cx.words + 2 * COOP_G, 1u);
            return false;
        }
        __builtin_amdgcn_s_sleep(RMX_COOP_SLEEP_C);
    }
    const double v = __hiloint2double((int)(unsigned)w1, (int)(unsigned)w0);
    dx = lane < 32 ? v : 0.0;
    gn2 = readlane_d(v, 32);
    code = (int)readlane_d(v, 33);
    T = readlane_d(v, 34);
    V = readlane_d(v, 35);
    return true;
}

// Flow control of the narrow rounds (newton_pair): there the decision words are not exchanged, so nothing keeps member 0 from
// publishing record s + 1 over a record s that a late member - one that has only just picked the rollout up - has not read yet.
// Every member therefore notes the last record it has handled in its own word (cx.words[COOP_ACK + m], relaxed), and member 0 looks
// at the nine words before it overwrites the record: all of them at s - 1 or beyond.  (In a wide round the exchange of the decision
// words itself is that guarantee: nobody posts its word before it has read the previous record, and the gather waits for every word.)
constexpr int COOP_ACK = 22;                     // cx.words[22 .. 31]
__device__ __forceinline__ void coop_ack(const CoopCtx& cx, const int lane, const unsigned seq) {
    if (lane == 0) coop_store(cx.words + COOP_ACK + cx.member, seq);
}
// first: this lane's acknowledgement word as read a while ago (coop_ack_peek, issued before the solve so that the round trip of the
// load hides behind it); only if that reading does not settle it is the word polled
__device__ __forceinline__ unsigned coop_ack_peek(const CoopCtx& cx, const int lane) {
    return (lane < COOP_G && lane != cx.member) ? coop_load(cx.words + COOP_ACK + lane) : 0xffffffffu;
}
__device__ __forceinline__ bool coop_wait_acks(const CoopCtx& cx, const int lane, const unsigned seq, const unsigned first) {
    const bool idle = !(lane < COOP_G && lane != cx.member);
    if (__all(idle || (int)(first - seq) >= 0)) return true;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (true) {
        unsigned w = seq;
        if (!idle) w = coop_load(cx.words + COOP_ACK + lane);
        if (__all((int)(w - seq) >= 0)) return true;
        unsigned ab = 0u;
        if (lane == 0) ab = coop_load(cx.words + 2 * COOP_G);
        if (__any(ab != 0u) || __builtin_amdgcn_s_memtime() - t0 > cx.ticks) {
            if (lane == 0) coop_store(cx.words + 2 * COOP_G, 1u);
            return false;
        }
        __builtin_amdgcn_s_sleep(2);
    }
}

// newton (driverRedMaxBDF1.m:94-157) for a serial chain of <= 32 nodes with the contact terms: decisions as newton_impl, see the
// header of this file.  pivot_all (wave-uniform): every solve with partial pivoting (lu_mode 1 / the pivot policy's hold).
template <bool COOP>
__device__ __forceinline__ double newton_pair(const DevModel& M, const DevOpts& o, double* sAcc, const int lane, double x, const double qA,
                                              const double qB, const double eta, NodeOut& last, int& iters, int& halvings, int& status,
                                              PivotPolicy& piv, const bool pivot_all, double& xlo, CoopCtx& cx, const CoopPub& pb,
                                              const double tau_add = 0.0) {
    constexpr int NP = 32;
    const double* cK = RMX_CONSTS(sAcc, M.n, NP);
    const double grav[3] = {M.grav[0], M.grav[1], M.grav[2]};
    const int halv_in = halvings;
    const bool hiH = lane >= 32;
    const double qAd = dup_lo(qA), qBd = dup_lo(qB);
    FrontState fs;
    fs.tau_add = tau_add;
    NodeOut e;
    double eT0 = 0.0, eV0 = 0.0;   // energies of the evaluation at x0 (what a stalled search leaves in `last`)
    double lo = 0.0, dx = 0.0, alpha = 1.0, f0 = 0.0, g0n2 = 0.0, x0 = x, lo0 = 0.0, gn2 = 0.0;
    int iter = 1, lsfail = 0, iterLs = 1;
    int mode = 0;                  // 0: evaluate x for the Hessian stage (first evaluation of the solve / before a pivoting re-solve);
                                   // 1: the next two (COOP, wide round: 2 COOP_G) trials of the line search
    bool redo = false;             // mode 0 is the re-evaluation before the pivoting solve of the same iteration
    last.g = last.eT = last.eV = 0.0;
#define RMX_COOP_GIVE_UP()              \
    {                                   \
        status |= 4 | ST_COOP_FAULT;    \
        xlo = lo0;                      \
        return x0;                      \
    }
    while (true) {
        // ---- A. the points of this evaluation
        double xl, lol;
        bool stall_a = false, stall_b = false;
        // COOP, narrow round (the last line search of this rollout ended within its first two trials - every ordinary Newton iteration
        // does): member 0 alone evaluates trials 1 and 2 and the group waits for its record; the decision words are not exchanged at all
        const bool narrow = COOP && mode == 1 && !cx.wide;
        const bool wide = COOP && mode == 1 && cx.wide;
        if (mode == 0) {
            xl = dup_lo(x);
            lol = dup_lo(lo);
        } else {
            const double x0d = dup_lo(x0), lo0d = dup_lo(lo0), dxd = dup_lo(dx);
            const double am = wide ? ldexp(alpha, -2 * cx.member) : alpha;      // (alpha is a power of two: exact)
            const double al = hiH ? 0.5 * am : am;
            two_sum(x0d, fma(al, dxd, lo0d), xl, lol);       // x + lo = x0 + (lo0 + alpha dx)
            lol *= o.comp;
            // alpha dx no longer changes the iterate in any DOF: this and every further halving re-evaluates g at x0 bit for bit, f == f0 is
            // never a strict decrease, the reference runs out its trials and keeps x0 (:132-138) - same outcome, without the evaluations
            const unsigned long long same = __ballot(xl == x0d && lol == lo0d);
            stall_a = (unsigned)same == 0xffffffffu;
            stall_b = (unsigned)(same >> 32) == 0xffffffffu;
        }
        // ---- B. the evaluation
        bool ta = false, tb = false;
        double ga2 = 0.0, gb2 = 0.0;
        const bool evaluates = !(mode == 1 && stall_a) && !(narrow && cx.member != 0);      // (a stalled: so are b and every later trial)
        if (evaluates) {
            RMX_PH_BEGIN(1)
            eval_front_pair<true>(M.n, cK, grav, lane, xl, ((xl - qAd) + lol) / eta, (xl - qBd) + lol, eta, e, fs, ta, tb);
            wave_sum_dual(e.g * e.g, ga2, gb2);
            RMX_PH_END(1)
        }
        // (COOP) the group's acknowledgement words, read NOW for the publish that may follow the Hessian stage and the solve
        unsigned ackw = 0u;
        if constexpr (COOP) {
            if (narrow) ackw = coop_ack_peek(cx, lane);
        }
        // ---- C. what can be decided before the Hessian stage, and whether the stage runs NOW, on which half's state
        //   mode 0: always (half a).
        //   a line search decided by this wavefront alone (not COOP; COOP narrow round, member 0): once the trial that ends the search
        //   is known and the solve does not end there.
        //   COOP wide round: SPECULATIVELY, between posting this member's decision word and collecting the group's - on the first of
        //   its two trials that could end the search.  If the walk over the group's words then names this member the winner, that is
        //   the accepted trial and the stage is already done; if not, the work is dropped (the member would have idled instead).
        int take = -1;                 // the trial, counted from iterLs, that ends the search
        bool stalled = false;
        bool conv = false, maxit = false, cut = false, park = false;     // the checks that follow a line search, for the accepted trial
        bool hess_now = mode == 0;
        int hess_half = 0;
        const bool local = mode == 1 && (!COOP || (narrow && cx.member == 0));
        if (local) {
            if (stall_a) stalled = true;
            else if (0.5 * ga2 < f0 || iterLs >= o.iterLsMax) take = 0;
            else if (stall_b) stalled = true;
            else if (0.5 * gb2 < f0 || iterLs + 1 >= o.iterLsMax) take = 1;
            if (take >= 0) {
                const double g2 = take ? gb2 : ga2;
                conv = sqrt(g2) < o.tol;
                maxit = iter >= o.iterMax;
                cut = o.lsFailLimit > 0 && lsfail + ((0.5 * g2 < f0) ? 0 : 1) >= o.lsFailLimit;
                park = !COOP && o.parkHalv > 0 && halvings + (iterLs + take - 1) - halv_in > o.parkHalv;
                hess_now = !(conv || maxit || cut || park);
                hess_half = take;
            }
        } else if (wide) {
            unsigned bits = (stall_a ? 1u : 0u) | (stall_b ? 2u : 0u);
            if (!stall_a) bits |= (0.5 * ga2 < f0 ? 4u : 0u) | (0.5 * gb2 < f0 ? 8u : 0u);
            coop_post(cx, lane, bits);
            const int ta_ix = iterLs + 2 * cx.member;                     // this member's trials, as line-search counts
            const bool cand_a = !stall_a && ((bits & 4u) || ta_ix >= o.iterLsMax);
            const bool cand_b = !stall_a && !stall_b && ((bits & 8u) || ta_ix + 1 >= o.iterLsMax);
            hess_now = cand_a || cand_b;
            hess_half = cand_a ? 0 : 1;
        }
        bool hdone = false;
        if (hess_now) {
            if (hess_half) front_take_hi(fs, e);
            fs.touched = hess_half ? tb : ta;
            double Hdummy[NP];
            RMX_PH_BEGIN(2)
            eval_hess<NP, false, true, false>(M, lane, fs, Hdummy, nullptr, sAcc, e.g);
            RMX_PH_END(2)
            hdone = true;
        }
        // ---- D. the line search's decision and what follows it
        bool need_solve = true;                            // this member runs the solve at the accepted point
        int take_pub = 0;                                  // (COOP) the accepted trial, as the winner's record names it
        if (mode == 1) {
            // (COOP) a record collected in the narrow round, used further down in place of the wait for the winner
            bool have_rec = false;
            double rT = 0.0, rV = 0.0, rdx = 0.0, rgn2 = 0.0;
            int rcode = 0;
            if constexpr (COOP) {
                if (narrow) {
                    if (cx.member == 0) {
                        if (stalled || take < 0) {         // nothing to hand over but the verdict: "stalled" / "the rest of the trials, everybody"
                            ++cx.rseq;
                            if (!coop_wait_acks(cx, lane, cx.rseq - 1, ackw)) RMX_COOP_GIVE_UP()
                            coop_publish(pb, cx.rseq, lane, 0.0, stalled ? COOP_CODE_STALL : COOP_CODE_WIDE, 0.0, 0.0, 0.0);
                            coop_ack(cx, lane, cx.rseq);
                        }
                    } else {
                        ++cx.rseq;
                        RMX_PH_BEGIN(5)
                        const bool cok = coop_collect(pb, cx, cx.rseq, lane, rgn2, rcode, rT, rV, rdx);
                        RMX_PH_END(5)
                        if (!cok) RMX_COOP_GIVE_UP()
                        have_rec = true;
                        coop_ack(cx, lane, cx.rseq);
                        const int what = rcode & 15;
                        if (what == COOP_CODE_STALL) stalled = true;
                        else if (what != COOP_CODE_WIDE) take = (rcode >> 4) & 31;
                    }
                } else {
                    unsigned word;
                    RMX_PH_BEGIN(4)
                    const bool xok = coop_gather(cx, lane, word, iterLs, o.iterLsMax, true);       // (every word: the exchange stays a barrier of the group;
                    // returning at the word that ends the search, with acknowledgements before every publish instead, measured no faster)
                    RMX_PH_END(4)
                    if (!xok) RMX_COOP_GIVE_UP()
#pragma unroll 1
                    for (int m = 0; m < COOP_G; ++m) {     // the reference's walk over the trials, in order
                        const unsigned w = (unsigned)__builtin_amdgcn_readlane((int)word, m);
                        if (w & 1u) { stalled = true; break; }
                        if ((w & 4u) || iterLs + 2 * m >= o.iterLsMax) { take = 2 * m; break; }
                        if (w & 2u) { stalled = true; break; }
                        if ((w & 8u) || iterLs + 2 * m + 1 >= o.iterLsMax) { take = 2 * m + 1; break; }
                    }
                }
            }
            bool mine = true;                              // the accepted trial was evaluated by this wavefront
            if constexpr (COOP) mine = take >= 0 && (narrow ? cx.member == 0 : (take >> 1) == cx.member);
            if (hdone && !(mine && !stalled)) {
                // a Hessian stage run ahead for a trial that did not end the search: the staging area goes back to the state the next
                // stage expects (row n of the accumulation scratch zero; see eval_hess)
                hdone = false;
                RMX_SYNC();
                if (lane < ACC_STRIDE) sAcc[M.n * ACC_STRIDE + lane] = 0.0;
                RMX_SYNC();
            }
            if (stalled) {
                // g is g(x0) again.  If it is not below tol the next Newton iteration is this one repeated exactly, and so on until
                // iter >= iterMax ("Newton did not converge", :150-153) with x unchanged: report that now.
                halvings += o.iterLsMax - 1;
                last.g = 0.0;
                last.eT = (COOP && lane != 0) ? 0.0 : eT0;     // (COOP: the energies travel as their sums, in lane 0)
                last.eV = (COOP && lane != 0) ? 0.0 : eV0;
                x = x0;
                lo = lo0;
                if (!(sqrt(g0n2) < o.tol)) status |= 2 | 8;
                break;
            }
            const int PER = wide ? 2 * COOP_G : 2;
            if (take < 0) {                                // none of this round's trials ends the search
                alpha = ldexp(alpha, -PER);
                iterLs += PER;
                if constexpr (COOP) cx.wide = true;        // the rest of this search, and the next ones, with the whole group
                continue;
            }
            iterLs += take;
            halvings += iterLs - 1;
            take_pub = take;
            if constexpr (COOP) {
                // every member: the accepted iterate from what all of them hold bit for bit
                two_sum(x0, fma(ldexp(alpha, -take), dx, lo0), x, lo);
                lo *= o.comp;
                cx.wide = iterLs > 2;                      // the next search starts with member 0 alone again once one ends within two trials
            } else {
                const double xs = (take & 1) ? take_hi(xl) : xl, ls = (take & 1) ? take_hi(lol) : lol;
                x = hiH ? x0 : xs;
                lo = hiH ? lo0 : ls;
            }
            int end_bits = -1;                             // >= 0: the solve ends here, with these status bits
            if (mine) {
                if (!hdone && (take & 1)) front_take_hi(fs, e);      // (hdone: the state is already where the stage worked on it)
                gn2 = (take & 1) ? gb2 : ga2;
                fs.touched = (take & 1) ? tb : ta;
                last.g = e.g;
                last.eT = hiH ? 0.0 : e.eT;
                last.eV = hiH ? 0.0 : e.eV;
                if constexpr (COOP) {                      // (the group passes the energies on as their sums)
                    last.eT = wave_sum(last.eT);
                    last.eV = wave_sum(last.eV);
                    last.eT = lane == 0 ? last.eT : 0.0;
                    last.eV = lane == 0 ? last.eV : 0.0;
                }
                if (!local) {                              // (a wide round's winner: the checks the deciding wavefront makes in C)
                    conv = sqrt(gn2) < o.tol;
                    maxit = iter >= o.iterMax;
                    cut = o.lsFailLimit > 0 && lsfail + ((0.5 * gn2 < f0) ? 0 : 1) >= o.lsFailLimit;
                }
                // a solve whose line searches keep running out their trials: hand the rollout over, at the start of this step, to
                // a cooperative group (see CoopCtx)
                if (park) {
                    status |= ST_PARK;
                    xlo = lo;
                    return x;
                }
                // the checks that follow a line search (:142-153; rmx_opts.ls_fail_limit: see newton_impl)
                if (conv) end_bits = 0;
                else if (maxit) end_bits = 2;              // "Newton did not converge"
                else {
                    lsfail += (0.5 * gn2 < f0) ? 0 : 1;
                    if (cut) end_bits = 2 | ST_LS_CUT;
                }
                if constexpr (COOP) {
                    if (end_bits >= 0) {
                        ++cx.rseq;
                        if (narrow && !coop_wait_acks(cx, lane, cx.rseq - 1, ackw)) RMX_COOP_GIVE_UP()
                        coop_publish(pb, cx.rseq, lane, gn2, COOP_CODE_END | (take << 4) | (end_bits << 12), readlane_d(last.eT, 0), readlane_d(last.eV, 0), 0.0);
                        coop_ack(cx, lane, cx.rseq);
                    }
                }
            } else {
                // (COOP only) wait for the winner: the end of the solve, the next direction, or "re-evaluate and pivot"
                double T = rT, V = rV, dxn = rdx;
                int code = rcode;
                if (have_rec) {
                    gn2 = rgn2;
                } else {
                    ++cx.rseq;
                    RMX_PH_BEGIN(5)
                    const bool cok = coop_collect(pb, cx, cx.rseq, lane, gn2, code, T, V, dxn);
                    RMX_PH_END(5)
                    if (!cok) RMX_COOP_GIVE_UP()
                    coop_ack(cx, lane, cx.rseq);
                }
                last.g = 0.0;
                last.eT = lane == 0 ? T : 0.0;
                last.eV = lane == 0 ? V : 0.0;
                const int what = code & 15;
                if (what == COOP_CODE_END) {
                    end_bits = code >> 12;
                } else {
                    lsfail += (0.5 * gn2 < f0) ? 0 : 1;
                    if (what == COOP_CODE_REDO) {          // the guarded solve tripped at the winner: everybody re-evaluates x and pivots
                        ++iter;
                        ++iters;
                        ++piv.streak;
                        status |= 16;
                        eT0 = readlane_d(last.eT, 0);
                        eV0 = readlane_d(last.eV, 0);
                        mode = 0;
                        redo = true;
                        continue;
                    }
                    need_solve = false;                    // COOP_CODE_DX
                    dx = dxn;
                }
            }
            if (end_bits >= 0) {
                status |= end_bits;
                if (hdone) {                               // (a stage run ahead at a point the solve ends at: see above)
                    RMX_SYNC();
                    if (lane < ACC_STRIDE) sAcc[M.n * ACC_STRIDE + lane] = 0.0;
                    RMX_SYNC();
                }
                break;
            }
            ++iter;
        } else {
            gn2 = ga2;
            if (!redo) {
                last.g = e.g;
                last.eT = hiH ? 0.0 : e.eT;
                last.eV = hiH ? 0.0 : e.eV;
            }
        }
        // ---- E. dx = -H\g at the accepted point (the Hessian stage has left H and -g in the staging area)
        if constexpr (COOP) {
            eT0 = readlane_d(wave_sum(last.eT), 0);
            eV0 = readlane_d(wave_sum(last.eV), 0);
            if (mode == 0) {                               // (keep the one form of `last` the cooperative solve uses: sums in lane 0)
                last.eT = lane == 0 ? eT0 : 0.0;
                last.eV = lane == 0 ? eV0 : 0.0;
            }
        } else {
            eT0 = last.eT;
            eV0 = last.eV;
        }
        if (!redo) ++iters;
        if (need_solve) {
            // (by construction the stage has run: mode 0 always, a locally decided search when the solve goes on, a wide round's winner
            // on its candidate trial)
            bool lu_ok = true;
            RMX_PH_BEGIN(3)
            if (pivot_all || redo) {
                double Hrow[NP];
                hess_rows_from_staging(M.n, lane, sAcc, Hrow);
                dx = lu_solve_neg<NP>(M.n, lane, Hrow, hiH ? 0.0 : e.g);
            } else {
                dx = lu_solve_neg_diag32(M.n, lane, sAcc, e.g, lu_ok);
            }
            RMX_PH_END(3)
            if (!lu_ok) {              // growth guard tripped: redo this solve with partial pivoting (H was destroyed in place)
                ++piv.streak;
                status |= 16;
                if constexpr (COOP) {
                    if (mode == 1) {
                        ++cx.rseq;
                        if (narrow && !coop_wait_acks(cx, lane, cx.rseq - 1, ackw)) RMX_COOP_GIVE_UP()
                        coop_publish(pb, cx.rseq, lane, gn2, COOP_CODE_REDO | (take_pub << 4), eT0, eV0, 0.0);
                        coop_ack(cx, lane, cx.rseq);
                    }
                }
                mode = 0;
                redo = true;
                continue;
            }
            if constexpr (COOP) {
                // the winner of the line search that led here hands the direction to the group (a first / redone evaluation was run
                // by every member: nothing to hand over)
                RMX_PH_BEGIN(6)
                if (mode == 1) {
                    ++cx.rseq;
                    if (narrow && !coop_wait_acks(cx, lane, cx.rseq - 1, ackw)) RMX_COOP_GIVE_UP()
                    coop_publish(pb, cx.rseq, lane, gn2, COOP_CODE_DX | (take_pub << 4), eT0, eV0, dx);
                    coop_ack(cx, lane, cx.rseq);
                }
                RMX_PH_END(6)
            }
        }
        if (!(pivot_all || redo)) piv.streak = 0;
        redo = false;
        const double dxn2 = wave_sum(hiH ? 0.0 : dx * dx);
        if (!(dxn2 == dxn2)) {         // NaN: give up on this trajectory instead of spinning to iterMax
            status |= 4;
            break;
        }
        if (sqrt(dxn2) > o.dxMax) {
            status |= 1;               // "Newton diverged" (:118-121): x is left at the last iterate
            break;
        }
        g0n2 = gn2;
        f0 = 0.5 * g0n2;
        x0 = x;
        lo0 = lo;
        alpha = 1.0;
        iterLs = 1;
        mode = 1;
    }
#undef RMX_COOP_GIVE_UP
    xlo = lo;
    return x;
}

}  // namespace rmx
