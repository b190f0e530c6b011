/*
 * redmax_tensorfree.c -- "Baseline B" of BASELINE.md / SURVEY.md §8(d): a tensor-free CPU implementation of the BDF1 step.
 *
 * TEST / BENCH INFRASTRUCTURE ONLY (same rules as redmax_oracle.c): loaded by tests/ and by bench.py's
 * cpu_baseline_tensor_free leg, never by the product.
 *
 * redmax_oracle.c restates the reference literally, including the nm x nr x nr tensors dJ/dq, dJdot/dq and the O(n^3)
 * ancestor loops of Joint.computeJacobian (matlab-diff/+redmax/Joint.m:534-612) - about 250x the flops the mathematics
 * needs.  A GPU/CPU ratio against that baseline mostly measures the reference's algorithmic waste.  This file is the other
 * baseline the survey asked for: the SAME algorithm the HIP kernels execute (world-frame recursive Newton-Euler with
 * analytic derivatives; tests/proto_worldframe.py is the executable derivation) in plain scalar C, with the
 * reference's Newton / line search (driverRedMaxBDF1.m:94-157) and an LU with partial pivoting, OpenMP over rollouts.
 *     g = M v - eta^2 f,  H = dg/dq   with qdot = (q - qA)/eta, v = q - qB     (evalBDF1, driverRedMaxBDF1.m:160-187)
 * Scope: trees of fixed / revolute / prismatic joints with joint springs, dampers and limits (Joint.computeForce,
 * Joint.m:437-487); no ground contact, no multi-DOF joints (the bench workloads have none).
 * Checked against the literal oracle in tests/test_oracle_tensorfree.py (g, H to 1e-11, rollouts to 1e-9).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "redmax_oracle.h"

typedef struct {
    int n, nr;
    int *parent, *type, *idx;
    double *LR, *Lp;     /* [n][9],[n][3]  parent body -> joint base frame (E0_ij(parent) E0_pj)            */
    double *RR, *Rp;     /* [n][9],[n][3]  E0_ji                                                            */
    double *axis;        /* [n][3] normalised                                                               */
    double *sb;          /* [n][6] A0_ij S  (Joint.m:508)                                                   */
    double *I4;          /* [n][4] I1 I2 I3 m                                                               */
    double *prm;         /* [n][8] tau stiffness damping qRest qLimL qLimU qLimK qLimD                      */
    double grav[3];
    /* workspace, per node */
    double *R, *p, *s, *phi, *xi, *beta, *S, *kd, *dd, *cu, *cl, *rl;
} tf_model;

static void cross3(const double* a, const double* b, double* c) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
static double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void mat3v(const double* R, const double* x, double* y) {
    for (int i = 0; i < 3; i++) y[i] = R[3 * i] * x[0] + R[3 * i + 1] * x[1] + R[3 * i + 2] * x[2];
}
static void mat3tv(const double* R, const double* x, double* y) {
    for (int i = 0; i < 3; i++) y[i] = R[i] * x[0] + R[3 + i] * x[1] + R[6 + i] * x[2];
}
static void sym3v(const double* S, const double* x, double* y) { /* xx xy xz yy yz zz */
    y[0] = S[0] * x[0] + S[1] * x[1] + S[2] * x[2];
    y[1] = S[1] * x[0] + S[3] * x[1] + S[4] * x[2];
    y[2] = S[2] * x[0] + S[4] * x[1] + S[5] * x[2];
}
static void mm3(const double* A, const double* B, double* C) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
/* column-major 4x4 -> R (row-major 3x3), p */
static void split_cm(const double* E, double* R, double* p) {
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) R[3 * i + j] = E[4 * j + i];
        p[i] = E[12 + i];
    }
}

static tf_model* tf_create(const orc_desc* d) {
    const int n = d->njoints;
    tf_model* m = (tf_model*)calloc(1, sizeof *m);
    m->n = n;
    m->parent = (int*)malloc(sizeof(int) * n);
    m->type = (int*)malloc(sizeof(int) * n);
    m->idx = (int*)malloc(sizeof(int) * n);
#define D(name, k) m->name = (double*)calloc((size_t)n * (k), sizeof(double))
    D(LR, 9); D(Lp, 3); D(RR, 9); D(Rp, 3); D(axis, 3); D(sb, 6); D(I4, 4); D(prm, 8);
    D(R, 9); D(p, 3); D(s, 6); D(phi, 6); D(xi, 6); D(beta, 6); D(S, 28); D(kd, 1); D(dd, 1); D(cu, 6); D(cl, 12); D(rl, 12);
#undef D
    int nr = 0;
    for (int j = n - 1; j >= 0; j--) m->idx[j] = d->type[j] != ORC_JOINT_FIXED ? nr++ : -1;   /* leaf-to-root, Scene.m:69-71 */
    m->nr = nr;
    for (int c = 0; c < 3; c++) m->grav[c] = d->grav[c];
    for (int j = 0; j < n; j++) {
        m->parent[j] = d->parent[j];
        m->type[j] = d->type[j];
        double Rpj[9], ppj[3];
        split_cm(d->E0_pj + 16 * j, Rpj, ppj);
        split_cm(d->E0_ji + 16 * j, m->RR + 9 * j, m->Rp + 3 * j);
        if (d->parent[j] >= 0) {   /* inv(E0_ji(parent)) * E0_pj */
            const double* Rq = m->RR + 9 * d->parent[j];
            const double* pq = m->Rp + 3 * d->parent[j];
            double Rt[9], dp[3];
            for (int i = 0; i < 3; i++)
                for (int k = 0; k < 3; k++) Rt[3 * i + k] = Rq[3 * k + i];
            mm3(Rt, Rpj, m->LR + 9 * j);
            for (int c = 0; c < 3; c++) dp[c] = ppj[c] - pq[c];
            mat3v(Rt, dp, m->Lp + 3 * j);
        } else {
            memcpy(m->LR + 9 * j, Rpj, sizeof Rpj);
            memcpy(m->Lp + 3 * j, ppj, sizeof ppj);
        }
        double a[3] = {d->axis[3 * j], d->axis[3 * j + 1], d->axis[3 * j + 2]};
        if (d->type[j] != ORC_JOINT_FIXED) {
            const double nn = sqrt(dot3(a, a));
            for (int c = 0; c < 3; c++) a[c] /= nn;
        }
        memcpy(m->axis + 3 * j, a, sizeof a);
        /* sb = Ad(E0_ij) S,  E0_ij = inv(E0_ji): R' , -R'p */
        double S6[6] = {0, 0, 0, 0, 0, 0};
        if (d->type[j] == ORC_JOINT_REVOLUTE) memcpy(S6, a, sizeof a);
        if (d->type[j] == ORC_JOINT_PRISMATIC) memcpy(S6 + 3, a, sizeof a);
        double w3[3], v3[3], pm[3], t3[3], cx[3];
        mat3tv(m->RR + 9 * j, S6, w3);
        mat3tv(m->RR + 9 * j, S6 + 3, v3);
        mat3tv(m->RR + 9 * j, m->Rp + 3 * j, t3);
        for (int c = 0; c < 3; c++) pm[c] = -t3[c];
        cross3(pm, w3, cx);
        for (int c = 0; c < 3; c++) {
            m->sb[6 * j + c] = w3[c];
            m->sb[6 * j + 3 + c] = v3[c] + cx[c];
        }
        for (int c = 0; c < 4; c++) m->I4[4 * j + c] = d->I_i[6 * j + c];
        double* P = m->prm + 8 * j;
        P[0] = d->tau ? d->tau[j] : 0.0;
        P[1] = d->stiffness ? d->stiffness[j] : 0.0;
        P[2] = d->damping ? d->damping[j] : 0.0;
        P[3] = d->q ? d->q[j] : 0.0;          /* qRest = q at init (Joint.m:157) */
        P[4] = d->qLimL ? d->qLimL[j] : -1e8;
        P[5] = d->qLimU ? d->qLimU[j] : 1e8;
        P[6] = d->qLimK ? d->qLimK[j] : 1e8;
        P[7] = d->qLimD ? d->qLimD[j] : 0.0;
    }
    return m;
}
static void tf_destroy(tf_model* m) {
    free(m->parent); free(m->type); free(m->idx);
    free(m->LR); free(m->Lp); free(m->RR); free(m->Rp); free(m->axis); free(m->sb); free(m->I4); free(m->prm);
    free(m->R); free(m->p); free(m->s); free(m->phi); free(m->xi); free(m->beta); free(m->S); free(m->kd); free(m->dd);
    free(m->cu); free(m->cl); free(m->rl);
    free(m);
}

/* g (and H, nr x nr column-major, when non-NULL) at x.  Two O(n) sweeps + one O(n * depth) Hessian fill. */
/* xlo (or NULL): low-order part of a compensated iterate x + xlo (|xlo| <= ulp(x)/2).  It enters where x enters LINEARLY with large
 * coefficients - v = x - qB (times M) and qdot = (x - qA)/eta (times D) - and nowhere else: the geometry is evaluated at x. */
static void tf_eval_lo(tf_model* m, const double* x, const double* xlo, const double* qA, const double* qB, double eta, double* g, double* H) {
    const int n = m->n, nr = m->nr;
    const double e2 = eta * eta;
    const double* gv = m->grav;
    /* ---- root -> leaves: transforms, screws, phi, xi, beta, body terms (Joint.update, Body.update, Body.computeMassGrav) */
    for (int j = 0; j < n; j++) {
        const int par = m->parent[j], k = m->idx[j];
        const double lo = (k >= 0 && xlo) ? xlo[k] : 0.0;
        const double q = k >= 0 ? x[k] : 0.0, qd = k >= 0 ? ((x[k] - qA[k]) + lo) / eta : 0.0, v = k >= 0 ? (x[k] - qB[k]) + lo : 0.0;
        double Q[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, qp[3] = {0, 0, 0};
        const double* a = m->axis + 3 * j;
        if (m->type[j] == ORC_JOINT_REVOLUTE) {      /* Rodrigues: a a' + cos (I - a a') + sin [a]   (se3.aaToMat) */
            const double sn = sin(q), cs = cos(q);
            const double ab[9] = {0, -a[2], a[1], a[2], 0, -a[0], -a[1], a[0], 0};
            for (int i = 0; i < 3; i++)
                for (int c = 0; c < 3; c++) Q[3 * i + c] = a[i] * a[c] + cs * ((i == c ? 1.0 : 0.0) - a[i] * a[c]) + sn * ab[3 * i + c];
        } else if (m->type[j] == ORC_JOINT_PRISMATIC) {
            for (int c = 0; c < 3; c++) qp[c] = a[c] * q;
        }
        /* T = L Q Rt ; E_w = E_w(parent) T */
        double T1[9], TR[9], t3[3], Tp[3];
        mm3(m->LR + 9 * j, Q, T1);
        mm3(T1, m->RR + 9 * j, TR);
        mat3v(T1, m->Rp + 3 * j, t3);
        mat3v(m->LR + 9 * j, qp, Tp);
        for (int c = 0; c < 3; c++) Tp[c] += t3[c] + m->Lp[3 * j + c];
        double* R = m->R + 9 * j;
        double* p = m->p + 3 * j;
        if (par >= 0) {
            mm3(m->R + 9 * par, TR, R);
            mat3v(m->R + 9 * par, Tp, p);
            for (int c = 0; c < 3; c++) p[c] += m->p[3 * par + c];
        } else {
            memcpy(R, TR, sizeof TR);
            memcpy(p, Tp, sizeof Tp);
        }
        double* s = m->s + 6 * j;
        mat3v(R, m->sb + 6 * j, s);
        mat3v(R, m->sb + 6 * j + 3, s + 3);
        cross3(p, s, t3);
        for (int c = 0; c < 3; c++) s[3 + c] += t3[c];
        double* phi = m->phi + 6 * j;
        double* xi = m->xi + 6 * j;
        double* beta = m->beta + 6 * j;
        for (int c = 0; c < 6; c++) phi[c] = (par >= 0 ? m->phi[6 * par + c] : 0.0) + s[c] * qd;
        cross3(phi, s, xi);
        cross3(phi + 3, s, xi + 3);
        cross3(phi, s + 3, t3);
        for (int c = 0; c < 3; c++) xi[3 + c] += t3[c];
        for (int c = 0; c < 6; c++) beta[c] = (par >= 0 ? m->beta[6 * par + c] : 0.0) + s[c] * v + e2 * qd * xi[c];
        /* body: world-frame inertia about the origin, momentum, wrench, B block */
        const double I1 = m->I4[4 * j], I2 = m->I4[4 * j + 1], I3 = m->I4[4 * j + 2], ms = m->I4[4 * j + 3];
        double mc[3], Ib[6];
        for (int c = 0; c < 3; c++) mc[c] = ms * p[c];
        const double cc = dot3(p, p);
        Ib[0] = I1 * R[0] * R[0] + I2 * R[1] * R[1] + I3 * R[2] * R[2] + ms * (cc - p[0] * p[0]);
        Ib[1] = I1 * R[0] * R[3] + I2 * R[1] * R[4] + I3 * R[2] * R[5] - ms * p[0] * p[1];
        Ib[2] = I1 * R[0] * R[6] + I2 * R[1] * R[7] + I3 * R[2] * R[8] - ms * p[0] * p[2];
        Ib[3] = I1 * R[3] * R[3] + I2 * R[4] * R[4] + I3 * R[5] * R[5] + ms * (cc - p[1] * p[1]);
        Ib[4] = I1 * R[3] * R[6] + I2 * R[4] * R[7] + I3 * R[5] * R[8] - ms * p[1] * p[2];
        Ib[5] = I1 * R[6] * R[6] + I2 * R[7] * R[7] + I3 * R[8] * R[8] + ms * (cc - p[2] * p[2]);
        double ht[3], hf[3], bt[3], bf[3], a3[3], b3[3], c3[3], fgt[3];
        sym3v(Ib, phi, ht); cross3(mc, phi + 3, t3);
        for (int c = 0; c < 3; c++) ht[c] += t3[c];
        cross3(mc, phi, t3);
        for (int c = 0; c < 3; c++) hf[c] = ms * phi[3 + c] - t3[c];
        sym3v(Ib, beta, bt); cross3(mc, beta + 3, t3);
        for (int c = 0; c < 3; c++) bt[c] += t3[c];
        cross3(mc, beta, t3);
        for (int c = 0; c < 3; c++) bf[c] = ms * beta[3 + c] - t3[c];
        cross3(phi, ht, a3); cross3(phi + 3, hf, b3); cross3(phi, hf, c3); cross3(mc, gv, fgt);
        double* S = m->S + 28 * j;
        for (int c = 0; c < 3; c++) {
            S[c] = bt[c] - e2 * (-a3[c] - b3[c] + fgt[c]);         /* w = I beta - eta^2 (ad(phi)' I phi + fgrav)   Body.m:102-109 */
            S[3 + c] = bf[c] - e2 * (-c3[c] + ms * gv[c]);
        }
        S[6] = ms;
        for (int c = 0; c < 3; c++) S[7 + c] = mc[c];
        for (int c = 0; c < 6; c++) S[10 + c] = Ib[c];
        {   /* TL = X + X' + [ht],  X = Ib [phi_w] + [mc][phi_v] */
            const double Ibf[9] = {Ib[0], Ib[1], Ib[2], Ib[1], Ib[3], Ib[4], Ib[2], Ib[4], Ib[5]};
            const double Om[9] = {0, -phi[2], phi[1], phi[2], 0, -phi[0], -phi[1], phi[0], 0};
            const double Vx[9] = {0, -phi[5], phi[4], phi[5], 0, -phi[3], -phi[4], phi[3], 0};
            const double Mc[9] = {0, -mc[2], mc[1], mc[2], 0, -mc[0], -mc[1], mc[0], 0};
            const double Ht[9] = {0, -ht[2], ht[1], ht[2], 0, -ht[0], -ht[1], ht[0], 0};
            double X1[9], X2[9];
            mm3(Ibf, Om, X1);
            mm3(Mc, Vx, X2);
            for (int i = 0; i < 3; i++)
                for (int c = 0; c < 3; c++) S[16 + 3 * i + c] = X1[3 * i + c] + X2[3 * i + c] + X1[3 * c + i] + X2[3 * c + i] + Ht[3 * i + c];
        }
        for (int c = 0; c < 3; c++) S[25 + c] = hf[c];
    }
    /* ---- leaves -> root: subtree sums (the J'(...) accumulations) */
    for (int j = n - 1; j > 0; j--) {
        const int par = m->parent[j];
        for (int c = 0; c < 28; c++) m->S[28 * par + c] += m->S[28 * j + c];
    }
    /* ---- residual and the per-node Hessian vectors */
    for (int j = 0; j < n; j++) {
        const int k = m->idx[j];
        if (k < 0) continue;
        const double q = x[k], qd = ((x[k] - qA[k]) + (xlo ? xlo[k] : 0.0)) / eta;
        const double* P = m->prm + 8 * j;
        const double hitL = q < P[4] ? 1.0 : 0.0, hitU = q > P[5] ? 1.0 : 0.0;
        const double fr = P[0] + P[1] * (P[3] - q) - P[2] * qd + hitL * (P[6] * (P[4] - q) - P[7] * qd) + hitU * (P[6] * (P[5] - q) - P[7] * qd);
        const double* s = m->s + 6 * j;
        const double* S = m->S + 28 * j;
        g[k] = dot3(s, S) + dot3(s + 3, S + 3) - e2 * fr;       /* evalBDF1 :180 */
        m->kd[j] = P[1] + (hitL + hitU) * P[6];                 /* -Kr, -Dr  (Joint.m:470-482) */
        m->dd[j] = P[2] + (hitL + hitU) * P[7];
    }
    if (!H) return;
    memset(H, 0, sizeof(double) * (size_t)nr * nr);
    for (int i = 0; i < n; i++) {
        const int ki = m->idx[i];
        if (ki < 0) continue;
        const double *sw = m->s + 6 * i, *sv = sw + 3, *phw = m->phi + 6 * i, *phv = phw + 3, *xiw = m->xi + 6 * i, *xiv = xiw + 3;
        const double *bw = m->beta + 6 * i, *bv = bw + 3;
        const double* S = m->S + 28 * i;
        const double *Wt = S, *Wf = S + 3, mS = S[6], *mcS = S + 7, *IbS = S + 10, *TL = S + 16, *hfS = S + 25;
        double zw[3], zv[3], t3[3], a3[3], b3[3], m1w[3], m1v[3], m2w[3];
        cross3(bw, sw, zw); cross3(bv, sw, zv); cross3(bw, sv, t3);
        for (int c = 0; c < 3; c++) zv[c] += t3[c];
        cross3(phw, xiw, a3); cross3(phv, xiw, b3); cross3(phw, xiv, t3);
        for (int c = 0; c < 3; c++) {
            zw[c] += e2 * a3[c];
            zv[c] += e2 * (b3[c] + t3[c]);
            m1w[c] = sw[c] + 2.0 * eta * xiw[c] + zw[c];
            m1v[c] = sv[c] + 2.0 * eta * xiv[c] + zv[c];
            m2w[c] = eta * sw[c] + e2 * xiw[c];
        }
        double yt[3], yf[3], gxs[3], kt[3];
        sym3v(IbS, m1w, yt); cross3(mcS, m1v, t3);
        for (int c = 0; c < 3; c++) yt[c] += t3[c];
        cross3(mcS, m1w, t3);
        for (int c = 0; c < 3; c++) yf[c] = mS * m1v[c] - t3[c];
        mat3v(TL, m2w, a3); cross3(hfS, m2w, b3); cross3(gv, sw, gxs); cross3(mcS, gxs, kt);
        for (int c = 0; c < 3; c++) {
            yt[c] -= a3[c] + e2 * kt[c];
            yf[c] -= 2.0 * b3[c] + e2 * mS * gxs[c];
        }
        double zt[3], zf[3];
        cross3(sw, Wt, a3); cross3(sv, Wf, b3); cross3(sw, Wf, zf);
        for (int c = 0; c < 3; c++) { zt[c] = -a3[c] - b3[c]; zf[c] = -zf[c]; }
        H[(size_t)ki * nr + ki] = dot3(sw, yt) + dot3(sv, yf) + eta * m->dd[i] + e2 * m->kd[i];
        double* cu = m->cu + 6 * i;      /* y - z : column i as seen from its strict ancestors */
        double* cl = m->cl + 12 * i;     /* m1, m2w, sw : column i as seen from its strict descendants */
        double* rl = m->rl + 12 * i;     /* r1, -r2w, -r3w : row i as a strict descendant */
        for (int c = 0; c < 3; c++) {
            cu[c] = yt[c] - zt[c]; cu[3 + c] = yf[c] - zf[c];
            cl[c] = m1w[c]; cl[3 + c] = m1v[c]; cl[6 + c] = m2w[c]; cl[9 + c] = sw[c];
        }
        sym3v(IbS, sw, rl); cross3(mcS, sv, t3);
        for (int c = 0; c < 3; c++) rl[c] += t3[c];
        cross3(mcS, sw, t3);
        for (int c = 0; c < 3; c++) rl[3 + c] = mS * sv[c] - t3[c];
        cross3(hfS, sv, b3);
        for (int c = 0; c < 3; c++) rl[6 + c] = -(TL[c] * sw[0] + TL[3 + c] * sw[1] + TL[6 + c] * sw[2] - 2.0 * b3[c]);
        cross3(gv, t3, a3); cross3(gv, sv, b3);
        for (int c = 0; c < 3; c++) rl[9 + c] = -e2 * (a3[c] - mS * b3[c]);
    }
    for (int i = 0; i < n; i++) {       /* every (ancestor a, node i) pair once: H(a,i) and H(i,a) */
        const int ki = m->idx[i];
        if (ki < 0) continue;
        for (int a = m->parent[i]; a >= 0; a = m->parent[a]) {
            const int ka = m->idx[a];
            if (ka < 0) continue;
            const double* sa = m->s + 6 * a;
            const double* cu = m->cu + 6 * i;
            double up = 0.0, lo = 0.0;
            for (int c = 0; c < 6; c++) up += sa[c] * cu[c];
            const double* rl = m->rl + 12 * i;
            const double* cl = m->cl + 12 * a;
            for (int c = 0; c < 12; c++) lo += rl[c] * cl[c];
            H[(size_t)ki * nr + ka] = up;       /* row a (ancestor), column i */
            H[(size_t)ka * nr + ki] = lo;       /* row i (descendant), column a */
        }
    }
}

static void tf_eval(tf_model* m, const double* x, const double* qA, const double* qB, double eta, double* g, double* H) {
    tf_eval_lo(m, x, NULL, qA, qB, eta, g, H);
}

/* dx = -H\g, LU with partial pivoting (MATLAB mldivide, driverRedMaxBDF1.m:117); H column-major, destroyed */
static void tf_solve_neg(int n, double* H, const double* g, double* dx) {
    for (int i = 0; i < n; i++) dx[i] = -g[i];
#define A(r, c) H[(size_t)(c) * n + (r)]
    for (int k = 0; k < n; k++) {
        int p = k;
        double mx = fabs(A(k, k));
        for (int r = k + 1; r < n; r++)
            if (fabs(A(r, k)) > mx) { mx = fabs(A(r, k)); p = r; }
        if (p != k) {
            for (int c = 0; c < n; c++) { const double t = A(k, c); A(k, c) = A(p, c); A(p, c) = t; }
            const double t = dx[k]; dx[k] = dx[p]; dx[p] = t;
        }
        const double inv = 1.0 / A(k, k);
        for (int r = k + 1; r < n; r++) A(r, k) *= inv;
        for (int c = k + 1; c < n; c++) {
            const double u = A(k, c);
            for (int r = k + 1; r < n; r++) A(r, c) -= A(r, k) * u;
        }
        for (int r = k + 1; r < n; r++) dx[r] -= A(r, k) * dx[k];
    }
    for (int k = n - 1; k >= 0; k--) {
        dx[k] /= A(k, k);
        for (int r = 0; r < k; r++) dx[r] -= A(r, k) * dx[k];
    }
#undef A
}

typedef struct { double *g, *H, *dx, *x0, *lo, *lo0; } tf_work;

/* newton (driverRedMaxBDF1.m:94-157), decision for decision.  comp != 0: the iterate is carried as the unevaluated sum x + lo
 * (the HIP kernels' default, rmx_opts.compensated): x + lo = x0 + (lo0 + alpha dx) exactly (TwoSum), lo enters the residual through
 * v and qdot only (tf_eval_lo); on return x is the iterate rounded to nearest and w->lo its low-order part. */
static void tf_newton(tf_model* m, tf_work* w, double* x, const double* qA, const double* qB, double eta, double tol, double dxMax,
                      int iterMax, int iterLsMax, int comp, int* iters, int* halvings, int* status) {
    const int nr = m->nr;
    int iter = 1;
    for (int i = 0; i < nr; i++) w->lo[i] = 0.0;
    while (1) {
        tf_eval_lo(m, x, w->lo, qA, qB, eta, w->g, w->H);
        ++*iters;
        tf_solve_neg(nr, w->H, w->g, w->dx);
        double dn = 0.0, f0 = 0.0;
        for (int i = 0; i < nr; i++) { dn += w->dx[i] * w->dx[i]; f0 += w->g[i] * w->g[i]; }
        if (sqrt(dn) > dxMax) { *status |= 1; break; }
        f0 *= 0.5;
        double alpha = 1.0, gn = 0.0;
        memcpy(w->x0, x, sizeof(double) * nr);
        memcpy(w->lo0, w->lo, sizeof(double) * nr);
        int iterLs = 1, stalled = 0;
        while (1) {
            int moved = 0;
            for (int i = 0; i < nr; i++) {
                const double a = w->x0[i], b = w->lo0[i] + alpha * w->dx[i];
                const double s = a + b, bb = s - a;
                x[i] = s;
                w->lo[i] = comp ? (a - (s - bb)) + (b - bb) : 0.0;
                moved |= (x[i] != w->x0[i]) || (w->lo[i] != w->lo0[i]);
            }
            if (!moved) {      /* the kernels' exact shortcut: every further halving re-evaluates g(x0) bit for bit, the reference runs
                                  out its trials, keeps x0 and repeats this very iteration until iterMax (rmx_device.h newton_impl) */
                stalled = 1;
                iterLs = iterLsMax;
                gn = 2.0 * f0;
                break;
            }
            tf_eval_lo(m, x, w->lo, qA, qB, eta, w->g, NULL);
            gn = 0.0;
            for (int i = 0; i < nr; i++) gn += w->g[i] * w->g[i];
            if (0.5 * gn < f0) break;
            if (iterLs >= iterLsMax) break;
            alpha *= 0.5;
            ++iterLs;
        }
        *halvings += iterLs - 1;
        if (stalled) {
            if (!(sqrt(gn) < tol)) *status |= 2 | 8;
            break;
        }
        if (sqrt(gn) < tol) break;
        if (iter >= iterMax) { *status |= 2; break; }
        ++iter;
    }
}

/* evalBDF1-style parity hook: g[nr], H[nr*nr] column-major or NULL */
void otf_eval(const orc_desc* d, const double* q, const double* qA, const double* qB, double eta, double* g, double* H) {
    tf_model* m = tf_create(d);
    tf_eval(m, q, qA, qB, eta, g, H);
    tf_destroy(m);
}
void otf_eval_lo(const orc_desc* d, const double* q, const double* qlo, const double* qA, const double* qB, double eta, double* g, double* H) {
    tf_model* m = tf_create(d);
    tf_eval_lo(m, q, qlo, qA, qB, eta, g, H);
    tf_destroy(m);
}
int otf_nr(const orc_desc* d) {
    int nr = 0;
    for (int j = 0; j < d->njoints; j++) nr += d->type[j] != ORC_JOINT_FIXED;
    return nr;
}

/* simLoop (driverRedMaxBDF1.m:57-91) for B rollouts, OpenMP over rollouts; q, qdot [B][nr] in/out.  iters / halvings /
 * status: per-rollout outputs [B] or NULL.  Returns the total Newton iteration count. */
long otf_batch_step_bdf1(const orc_desc* d, int B, double* q, double* qdot, double h, int nsteps, int nthreads, double tol,
                         double dxMax, int iterMaxPerDof, int iterLsMax, int compensated, int* iters, int* halvings, int* status) {
    long total = 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel reduction(+ : total)
    {
        tf_model* m = tf_create(d);
        const int nr = m->nr;
        tf_work w;
        w.g = (double*)malloc(sizeof(double) * (nr + 1));
        w.H = (double*)malloc(sizeof(double) * ((size_t)nr * nr + 1));
        w.dx = (double*)malloc(sizeof(double) * (nr + 1));
        w.x0 = (double*)malloc(sizeof(double) * (nr + 1));
        w.lo = (double*)calloc((size_t)nr + 1, sizeof(double));
        w.lo0 = (double*)calloc((size_t)nr + 1, sizeof(double));
        double* x = (double*)malloc(sizeof(double) * (nr + 1));
        double* xB = (double*)malloc(sizeof(double) * (nr + 1));
#pragma omp for schedule(dynamic, 1)
        for (int b = 0; b < B; b++) {
            double* qb = q + (size_t)b * nr;
            double* qdb = qdot + (size_t)b * nr;
            int it = 0, hv = 0, st = 0;
            for (int s = 0; s < nsteps; s++) {
                for (int i = 0; i < nr; i++) x[i] = xB[i] = qb[i] + h * qdb[i];      /* initial guess = q0 + h qdot0 (:70, :169) */
                tf_newton(m, &w, x, qb, xB, h, tol, dxMax, iterMaxPerDof * nr, iterLsMax, compensated, &it, &hv, &st);
                for (int i = 0; i < nr; i++) { qdb[i] = ((x[i] - qb[i]) + w.lo[i]) / h; qb[i] = x[i]; }
            }
            if (iters) iters[b] = it;
            if (halvings) halvings[b] = hv;
            if (status) status[b] = st;
            total += it;
        }
        free(w.g); free(w.H); free(w.dx); free(w.x0); free(w.lo); free(w.lo0); free(x); free(xB);
        tf_destroy(m);
    }
    return total;
}
