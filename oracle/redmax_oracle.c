/*
 * redmax_oracle.c -- CPU restatement (plain C, fp64, single trajectory) of the
 * sueda/redmax matlab-diff BDF1/BDF2 forward-dynamics path.
 *
 * TEST INFRASTRUCTURE ONLY (see redmax_oracle.h).  Literal restatement: same data
 * flow as the reference .m files, including the dense dJ/dq, dJdot/dq tensors and the
 * O(n^3) ancestor loops of Joint.computeJacobian.  The only liberty taken is that the
 * block-diagonal maximal matrices Mm/Km/Dm are multiplied block-wise and exact-zero
 * tensor entries are skipped (adds of exact zeros are omitted; values are unchanged).
 *
 * Reference files restated (paths relative to /root/reference/):
 *   matlab-diff/se3.m, matlab-diff/+redmax/{Scene,Joint,JointRevolute,JointPrismatic,
 *   JointFixed,Body,BodyCuboid}.m, matlab-diff/driverRedMaxBDF1.m, driverRedMaxBDF2.m,
 *   matlab-simple/testRedMax.m (euler) with matlab-simple/+redmax/{Joint,Body}.m.
 *
 * Internal matrices are row-major C arrays; everything crossing the API is
 * column-major (MATLAB layout).
 */
#include "redmax_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef double m4[4][4];
typedef double m6[6][6];
typedef double v6[6];

/* ------------------------------------------------------------------ se3.m */

static void m4_eye(m4 E) { memset(E, 0, sizeof(m4)); for (int i = 0; i < 4; i++) E[i][i] = 1.0; }
static void m4_copy(m4 D, const m4 S) { memcpy(D, S, sizeof(m4)); }
static void m4_mul(m4 C, const m4 A, const m4 B) {
    m4 T;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) {
        double s = 0; for (int k = 0; k < 4; k++) s += A[i][k] * B[k][j]; T[i][j] = s;
    }
    memcpy(C, T, sizeof(m4));
}
static void m6_zero(m6 A) { memset(A, 0, sizeof(m6)); }
static void m6_mul(m6 C, const m6 A, const m6 B) {
    m6 T;
    for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) {
        double s = 0; for (int k = 0; k < 6; k++) s += A[i][k] * B[k][j]; T[i][j] = s;
    }
    memcpy(C, T, sizeof(m6));
}
static void m6_mulv(v6 y, const m6 A, const v6 x) {
    v6 t;
    for (int i = 0; i < 6; i++) { double s = 0; for (int k = 0; k < 6; k++) s += A[i][k] * x[k]; t[i] = s; }
    memcpy(y, t, sizeof(v6));
}

/* se3.inv  (se3.m:11-16) */
static void se3_inv(m4 Ei, const m4 E) {
    m4 T; m4_eye(T);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) T[i][j] = E[j][i];
    for (int i = 0; i < 3; i++) {
        double s = 0; for (int k = 0; k < 3; k++) s += E[k][i] * E[k][3];
        T[i][3] = -s;
    }
    memcpy(Ei, T, sizeof(m4));
}
/* se3.brac, 3-vector form (se3.m:89-98) */
static void se3_brac3(double S[3][3], const double x[3]) {
    S[0][0] = 0;     S[0][1] = -x[2]; S[0][2] = x[1];
    S[1][0] = x[2];  S[1][1] = 0;     S[1][2] = -x[0];
    S[2][0] = -x[1]; S[2][1] = x[0];  S[2][2] = 0;
}
/* se3.Ad (se3.m:44-52) */
static void se3_Ad(m6 A, const m4 E) {
    double pb[3][3], p[3] = { E[0][3], E[1][3], E[2][3] };
    m6_zero(A);
    se3_brac3(pb, p);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        A[i][j] = E[i][j];
        A[3 + i][3 + j] = E[i][j];
        double s = 0; for (int k = 0; k < 3; k++) s += pb[i][k] * E[k][j];
        A[3 + i][j] = s;
    }
}
/* se3.ad, 6-vector form (se3.m:55-69) */
static void se3_ad(m6 a, const v6 phi) {
    double W[3][3], Vb[3][3];
    m6_zero(a);
    se3_brac3(W, phi); se3_brac3(Vb, phi + 3);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        a[i][j] = W[i][j]; a[3 + i][j] = Vb[i][j]; a[3 + i][3 + j] = W[i][j];
    }
}
/* se3.Addot (se3.m:72-86) */
static void se3_Addot(m6 dA, const m4 E, const v6 phi) {
    double wb[3][3], vb[3][3], pb[3][3], Rw[3][3];
    double p[3] = { E[0][3], E[1][3], E[2][3] };
    m6_zero(dA);
    se3_brac3(wb, phi); se3_brac3(vb, phi + 3); se3_brac3(pb, p);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        double s = 0; for (int k = 0; k < 3; k++) s += E[i][k] * wb[k][j]; Rw[i][j] = s;
    }
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        dA[i][j] = Rw[i][j]; dA[3 + i][3 + j] = Rw[i][j];
        double s = 0;
        for (int k = 0; k < 3; k++) s += E[i][k] * vb[k][j] + pb[i][k] * Rw[k][j];
        dA[3 + i][j] = s;
    }
}
/* se3.aaToMat (se3.m:111-176), THRESH = 1e-9 (se3.m:5) */
static void se3_aaToMat(double R[3][3], const double axis[3], double angle) {
    const double THRESH = 1e-9;
    double ax = axis[0], ay = axis[1], az = axis[2];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[i][j] = (i == j);
    double mag = sqrt(ax * ax + ay * ay + az * az);
    if (mag > THRESH) {
        mag = 1.0 / mag; ax *= mag; ay *= mag; az *= mag;
        if (fabs(ax) < THRESH && fabs(ay) < THRESH) {          /* about Z */
            if (az < 0) angle = -angle;
            double s = sin(angle), c = cos(angle);
            R[0][0] = c; R[0][1] = -s; R[1][0] = s; R[1][1] = c;
        } else if (fabs(ay) < THRESH && fabs(az) < THRESH) {   /* about X */
            if (ax < 0) angle = -angle;
            double s = sin(angle), c = cos(angle);
            R[1][1] = c; R[1][2] = -s; R[2][1] = s; R[2][2] = c;
        } else if (fabs(az) < THRESH && fabs(ax) < THRESH) {   /* about Y */
            if (ay < 0) angle = -angle;
            double s = sin(angle), c = cos(angle);
            R[0][0] = c; R[0][2] = s; R[2][0] = -s; R[2][2] = c;
        } else {                                               /* general */
            double s = sin(angle), c = cos(angle), t = 1.0 - c;
            double xz = ax * az, xy = ax * ay, yz = ay * az;
            R[0][0] = t * ax * ax + c;  R[0][1] = t * xy - s * az;  R[0][2] = t * xz + s * ay;
            R[1][0] = t * xy + s * az;  R[1][1] = t * ay * ay + c;  R[1][2] = t * yz - s * ax;
            R[2][0] = t * xz - s * ay;  R[2][1] = t * yz + s * ax;  R[2][2] = t * az * az + c;
        }
    }
}

/* ------------------------------------------------------------ scene state */

typedef struct {
    /* constants */
    int parent, type, ndof, idxR, idxM;
    double axis[3];
    int has_E0_pj;
    m4 E0_pj, E0_jp;            /* Joint.setJointTransform Joint.m:95-99 */
    m4 E0_ji, E0_ij; m6 A0_ij;  /* Body.setBodyTransform  Body.m:46-51  */
    double I_i[6];
    double tau, stiffness, damping, qRest, qLimL, qLimU, qLimK, qLimD;
    /* state */
    double q, qdot, q0, qdot0, q1, qdot1;
    /* Joint.update outputs */
    m4 Q, invQ, E_pj, E_jp, E_wj;
    m6 A, Adot, dAdq, dAdotdq, invA, A_jp;
    v6 S, V;
    /* Body.update outputs */
    m4 E_wi, E_iw; v6 phi;
    /* matlab-simple extras */
    m6 Ad_wi, Ad_iw, Ad_ip, Addot_wi;
} onode;

struct orc_scene {
    int n, nr, nm;
    int normalize_axis;
    double grav[3];
    onode* nd;
    double* qInit; double* qdotInit;
    double T0, V0;
    /* workspaces */
    double *J, *Jdot, *dJdq, *dJdotdq;   /* col-major nm x nr (x nr) */
    double *Mm, *Km, *Dm;                /* n blocks of 36 (row-major 6x6), in idxM/6 order */
    double *fm;                          /* nm */
    double *fr, *Kr, *Dr;                /* nr, nr (diagonals)      */
    /* ForceGroundCuboid (one per flagged body, shared parameters) */
    int* contact;                        /* [n] flag */
    double* sides;                       /* [n][3]   */
    struct ogf { m4 E; double kn, kt, mu, kd; } *gf;      /* [n] the ground frame and constants of each body's ForceGroundCuboid */
    /* further ForceGroundCuboid objects on bodies that already carry one: the reference keeps its forces in a linked list
     * (Force.m:26-56, Scene.m:87-89) and nothing limits a body to one (a floor and a wall) */
    int nxf; int* xf_body; struct ogf* xf;
    /* JointSpherical groups (three consecutive revolute nodes each) and their Euler charts */
    int nsph; int* sph_first; int* sph_chart; int* sph_chart1;
};

#define JX(s, r, c) ((s)->J[(size_t)(c) * (s)->nm + (r)])
#define JDX(s, r, c) ((s)->Jdot[(size_t)(c) * (s)->nm + (r)])
#define T3(p, s, r, c, k) ((p)[((size_t)(k) * (s)->nr + (c)) * (s)->nm + (r)])

static void cm16_to_m4(m4 E, const double* cm) { for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) E[i][j] = cm[j * 4 + i]; }

int orc_nr(const orc_scene* s) { return s->nr; }
int orc_nm(const orc_scene* s) { return s->nm; }
void orc_idxR(const orc_scene* s, int* idx) { for (int i = 0; i < s->n; i++) idx[i] = s->nd[i].ndof ? s->nd[i].idxR : -1; }

/* JointRevolute.update_ (JointRevolute.m:29-53), JointPrismatic.update_ (JointPrismatic.m:28-42),
 * JointFixed (no update_: Q=I, A=I, Adot=0). */
static void joint_update_(orc_scene* s, onode* j) {
    (void)s;
    m4_eye(j->Q);
    m6_zero(j->A); for (int i = 0; i < 6; i++) j->A[i][i] = 1.0;
    m6_zero(j->Adot); m6_zero(j->dAdq); m6_zero(j->dAdotdq);
    for (int i = 0; i < 6; i++) j->S[i] = 0.0;
    if (j->type == ORC_JOINT_REVOLUTE) {
        double R[3][3], ab[3][3], dRdq[3][3], d2R[3][3];
        se3_aaToMat(R, j->axis, j->q);
        for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) j->Q[a][b] = R[a][b];
        se3_Ad(j->A, j->Q);
        for (int a = 0; a < 3; a++) j->S[a] = j->axis[a];
        se3_brac3(ab, j->axis);
        for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) {
            double t = 0; for (int k = 0; k < 3; k++) t += R[a][k] * ab[k][b]; dRdq[a][b] = t;
        }
        for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) {
            double t = 0; for (int k = 0; k < 3; k++) t += dRdq[a][k] * ab[k][b]; d2R[a][b] = t;
        }
        for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) {
            double Rdot = dRdq[a][b] * j->qdot;
            j->Adot[a][b] = Rdot; j->Adot[3 + a][3 + b] = Rdot;
            j->dAdq[a][b] = dRdq[a][b]; j->dAdq[3 + a][3 + b] = dRdq[a][b];
            double t = d2R[a][b] * j->qdot;
            j->dAdotdq[a][b] = t; j->dAdotdq[3 + a][3 + b] = t;
        }
    } else if (j->type == ORC_JOINT_PRISMATIC) {
        double ab[3][3];
        for (int a = 0; a < 3; a++) j->Q[a][3] = j->axis[a] * j->q;
        se3_Ad(j->A, j->Q);
        for (int a = 0; a < 3; a++) j->S[3 + a] = j->axis[a];
        se3_brac3(ab, j->axis);
        for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) {
            j->Adot[3 + a][b] = ab[a][b] * j->qdot;
            j->dAdq[3 + a][b] = ab[a][b];
        }
    }
}

/* Joint.update (Joint.m:382-434) + Body.update (Body.m:70-80); the extra Ad_* fields are the
 * matlab-simple Body.update additions (matlab-simple/+redmax/Body.m:89-104). */
static void scene_update(orc_scene* s) {
    for (int i = 0; i < s->n; i++) {
        onode* j = &s->nd[i];
        joint_update_(s, j);
        se3_inv(j->invQ, j->Q);
        se3_Ad(j->invA, j->invQ);
        if (!j->has_E0_pj) m4_copy(j->E_pj, j->Q); else m4_mul(j->E_pj, j->E0_pj, j->Q);
        se3_inv(j->E_jp, j->E_pj);
        se3_Ad(j->A_jp, j->E_jp);
        if (j->parent < 0) m4_copy(j->E_wj, j->E_pj);
        else m4_mul(j->E_wj, s->nd[j->parent].E_wj, j->E_pj);
        for (int a = 0; a < 6; a++) j->V[a] = j->ndof ? j->S[a] * j->qdot : 0.0;
        if (j->parent >= 0) {
            v6 t; m6_mulv(t, j->A_jp, s->nd[j->parent].V);
            for (int a = 0; a < 6; a++) j->V[a] += t[a];
        }
        /* Body.update */
        m4_mul(j->E_wi, j->E_wj, j->E0_ji);
        se3_inv(j->E_iw, j->E_wi);
        m6_mulv(j->phi, j->A0_ij, j->V);
        /* matlab-simple extras */
        se3_Ad(j->Ad_wi, j->E_wi);
        se3_Ad(j->Ad_iw, j->E_iw);
        if (j->parent >= 0) { m4 E_ip; m4_mul(E_ip, j->E_iw, s->nd[j->parent].E_wi); se3_Ad(j->Ad_ip, E_ip); }
        se3_Addot(j->Addot_wi, j->E_wi, j->phi);
    }
}

static double ground_contact_energy(const orc_scene* s);

/* Joint.computeEnergies (Joint.m:616-637) + Body.computeEnergies (Body.m:167-173) */
void orc_energy(const orc_scene* s, double* T, double* V) {
    double t = 0, v = 0;
    for (int i = 0; i < s->n; i++) {
        const onode* j = &s->nd[i];
        double tt = 0; for (int a = 0; a < 6; a++) tt += j->phi[a] * j->I_i[a] * j->phi[a];
        t += 0.5 * tt;
        double gp = 0; for (int a = 0; a < 3; a++) gp += s->grav[a] * j->E_wi[a][3];
        v -= j->I_i[5] * gp;
        if (j->ndof) {
            double dq = j->q - j->qRest;
            v += 0.5 * j->stiffness * (dq * dq);
            double dqL = (j->q < j->qLimL) ? (j->qLimL - j->q) : 0.0;
            double dqU = (j->q > j->qLimU) ? (j->qLimU - j->q) : 0.0;
            v += 0.5 * j->qLimK * (dqL * dqL + dqU * dqU);
        }
    }
    *T = t; *V = v + ground_contact_energy(s);   /* forces{1}.computeEnergy(V)  Scene.m:127,156 */
}

void orc_get_state(const orc_scene* s, double* q, double* qdot) {
    for (int i = 0; i < s->n; i++) if (s->nd[i].ndof) {
        if (q) q[s->nd[i].idxR] = s->nd[i].q;
        if (qdot) qdot[s->nd[i].idxR] = s->nd[i].qdot;
    }
}
static void set_q(orc_scene* s, const double* q, const double* qdot) {
    for (int i = 0; i < s->n; i++) if (s->nd[i].ndof) {
        if (q) s->nd[i].q = q[s->nd[i].idxR];
        if (qdot) s->nd[i].qdot = qdot[s->nd[i].idxR];
    }
}
void orc_set_state(orc_scene* s, const double* q, const double* qdot) { set_q(s, q, qdot); scene_update(s); }
void orc_set_qrest(orc_scene* s, const double* qrest) {
    for (int i = 0; i < s->n; i++) if (s->nd[i].ndof) s->nd[i].qRest = qrest[s->nd[i].idxR];
}

/* Overrides the reduced numbering: idx[i] for every joint with a DOF (a permutation of 0..nr-1).  Used when a multi-DOF
 * joint of the reference (JointPlanar / Translational / Universal / Free2D, whose idxR = nr + (1:ndof), Joint.m:152) is
 * restated as a chain of 1-DOF joints: the chain's DOFs then keep the reference's order q(1), q(2), ... */
int orc_set_idxR(orc_scene* s, const int* idx) {
    char* seen = (char*)calloc((size_t)(s->nr > 0 ? s->nr : 1), 1);
    for (int i = 0; i < s->n; i++) if (s->nd[i].ndof) {
        if (idx[i] < 0 || idx[i] >= s->nr || seen[idx[i]]) { free(seen); return -1; }
        seen[idx[i]] = 1;
    }
    free(seen);
    for (int i = 0; i < s->n; i++) if (s->nd[i].ndof) s->nd[i].idxR = idx[i];
    orc_get_state(s, s->qInit, s->qdotInit);
    orc_reset(s);
    return 0;
}

/* Scene.reset (Scene.m:122-131) */
void orc_reset(orc_scene* s) {
    set_q(s, s->qInit, s->qdotInit);
    scene_update(s);   /* the reference relies on init()'s update(); restated explicitly so reset is re-usable */
    orc_energy(s, &s->T0, &s->V0);
}

/* Scene.init (Scene.m:59-119), Joint.countDofs (Joint.m:149-158), Body.countDofs (Body.m:54-60) */
orc_scene* orc_create(const orc_desc* d) {
    orc_scene* s = (orc_scene*)calloc(1, sizeof(orc_scene));
    int n = d->njoints;
    s->n = n; s->normalize_axis = d->normalize_axis;
    memcpy(s->grav, d->grav, sizeof(s->grav));
    s->nd = (onode*)calloc((size_t)n, sizeof(onode));
    for (int i = 0; i < n; i++) {
        onode* j = &s->nd[i];
        j->parent = d->parent[i]; j->type = d->type[i];
        j->ndof = (j->type == ORC_JOINT_FIXED) ? 0 : 1;
        double ax[3] = { d->axis[3 * i], d->axis[3 * i + 1], d->axis[3 * i + 2] };
        if (d->normalize_axis && j->ndof) {    /* JointRevolute.m:14, JointPrismatic.m:15 */
            double nn = sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
            for (int a = 0; a < 3; a++) ax[a] /= nn;
        }
        memcpy(j->axis, ax, sizeof(ax));
        j->has_E0_pj = (d->E0_pj != NULL);
        if (d->E0_pj) cm16_to_m4(j->E0_pj, d->E0_pj + 16 * i); else m4_eye(j->E0_pj);
        se3_inv(j->E0_jp, j->E0_pj);
        cm16_to_m4(j->E0_ji, d->E0_ji + 16 * i);
        se3_inv(j->E0_ij, j->E0_ji);
        se3_Ad(j->A0_ij, j->E0_ij);
        memcpy(j->I_i, d->I_i + 6 * i, 6 * sizeof(double));
        j->tau = d->tau ? d->tau[i] : 0.0;
        j->stiffness = d->stiffness ? d->stiffness[i] : 0.0;
        j->damping = d->damping ? d->damping[i] : 0.0;
        j->qLimL = d->qLimL ? d->qLimL[i] : -1e8;     /* Joint.m:77-80 defaults */
        j->qLimU = d->qLimU ? d->qLimU[i] : 1e8;
        j->qLimK = d->qLimK ? d->qLimK[i] : 1e8;
        j->qLimD = d->qLimD ? d->qLimD[i] : 0.0;
        j->q = (j->ndof && d->q) ? d->q[i] : 0.0;
        j->qdot = (j->ndof && d->qdot) ? d->qdot[i] : 0.0;
    }
    /* leaf-to-root counting: Scene.m:69-71 */
    int nr = 0, nm = 0;
    for (int i = n - 1; i >= 0; i--) {
        onode* j = &s->nd[i];
        j->idxR = nr; nr += j->ndof;
        j->idxM = nm; nm += 6;
        j->qRest = j->q;                 /* Joint.m:157 */
    }
    s->nr = nr; s->nm = nm;
    for (int i = 0; i < n; i++) {        /* Scene.m:92-97 */
        onode* j = &s->nd[i];
        j->q0 = j->q1 = j->q; j->qdot0 = j->qdot1 = j->qdot;
    }
    s->qInit = (double*)calloc((size_t)(nr > 0 ? nr : 1), sizeof(double));
    s->qdotInit = (double*)calloc((size_t)(nr > 0 ? nr : 1), sizeof(double));
    size_t snr = (size_t)(nr > 0 ? nr : 1);
    s->J = (double*)calloc((size_t)nm * snr, sizeof(double));
    s->Jdot = (double*)calloc((size_t)nm * snr, sizeof(double));
    s->dJdq = (double*)calloc((size_t)nm * snr * snr, sizeof(double));
    s->dJdotdq = (double*)calloc((size_t)nm * snr * snr, sizeof(double));
    s->Mm = (double*)calloc((size_t)n * 36, sizeof(double));
    s->Km = (double*)calloc((size_t)n * 36, sizeof(double));
    s->Dm = (double*)calloc((size_t)n * 36, sizeof(double));
    s->fm = (double*)calloc((size_t)nm, sizeof(double));
    s->fr = (double*)calloc(snr, sizeof(double));
    s->Kr = (double*)calloc(snr, sizeof(double));
    s->Dr = (double*)calloc(snr, sizeof(double));
    scene_update(s);                               /* Scene.m:100 */
    orc_get_state(s, s->qInit, s->qdotInit);       /* Scene.m:101 */
    orc_reset(s);                                  /* Scene.m:118 */
    return s;
}

void orc_destroy(orc_scene* s) {
    if (!s) return;
    free(s->nd); free(s->qInit); free(s->qdotInit);
    free(s->J); free(s->Jdot); free(s->dJdq); free(s->dJdotdq);
    free(s->Mm); free(s->Km); free(s->Dm); free(s->fm); free(s->fr); free(s->Kr); free(s->Dr);
    free(s->contact); free(s->sides); free(s->gf); free(s->xf_body); free(s->xf);
    free(s->sph_first); free(s->sph_chart); free(s->sph_chart1);
    free(s);
}

/* -------------------------------------------------- Joint.computeJacobian */

static void blk_mul_col(double* out, const m6 A, const double* in) {   /* out(6) = A*in(6) */
    for (int r = 0; r < 6; r++) { double t = 0; for (int k = 0; k < 6; k++) t += A[r][k] * in[k]; out[r] = t; }
}

/* Joint.computeJacobian (Joint.m:490-613). deriv=0: the 2-output O(n^2) branch (:493-532);
 * deriv=1: the 4-output O(n^3) branch (:533-612). Traversal = listing order (next links). */
static void compute_jacobian(orc_scene* s, int deriv) {
    const int nm = s->nm, nr = s->nr;
    memset(s->J, 0, sizeof(double) * (size_t)nm * nr);
    memset(s->Jdot, 0, sizeof(double) * (size_t)nm * nr);
    if (deriv) {
        memset(s->dJdq, 0, sizeof(double) * (size_t)nm * nr * nr);
        memset(s->dJdotdq, 0, sizeof(double) * (size_t)nm * nr * nr);
    }
    for (int i = 0; i < s->n; i++) {
        onode* ji = &s->nd[i];
        const int rI = ji->idxM;
        if (ji->ndof) {
            /* J(idxmI,idxrI) = A0_BiJi*S ; Jdot(idxmI,idxrI) = A0_BiJi*Sdot (=0)   Joint.m:508-509,553-554 */
            double col[6]; blk_mul_col(col, ji->A0_ij, ji->S);
            for (int r = 0; r < 6; r++) { JX(s, rI + r, ji->idxR) = col[r]; JDX(s, rI + r, ji->idxR) = 0.0; }
            /* dJdq(idxmI,idxrI,idxrI(ii)) = A0_BiJi*dSdq = 0 for revolute/prismatic     Joint.m:555-558 */
        }
        if (ji->parent < 0) continue;
        onode* jp = &s->nd[ji->parent];
        const int rP = jp->idxM;
        m4 E0_JiBp, E_BiBp, T1, T2;
        m6 A_BiBp, Aleft, Aright, Adot_BiBp, dAdq_BiBp, dAdotdq_BiBp, T6;
        m4_mul(E0_JiBp, ji->E0_jp, jp->E0_ji);          /* Joint.m:512-514 */
        m4_mul(T1, ji->E0_ij, ji->invQ);                /* E0_BiJi*invQ    */
        m4_mul(E_BiBp, T1, E0_JiBp);                    /* Joint.m:515     */
        se3_Ad(A_BiBp, E_BiBp);
        se3_Ad(Aleft, T1);
        for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) Aleft[a][b] = -Aleft[a][b];   /* :517 */
        m4_mul(T2, ji->invQ, E0_JiBp);
        se3_Ad(Aright, T2);                             /* :518 */
        m6_mul(T6, Aleft, ji->Adot); m6_mul(Adot_BiBp, T6, Aright);   /* :519 */
        if (deriv && ji->ndof) {                        /* Joint.m:573-580 (ndof==1) */
            m6 tmp1, tmp2, U;
            m6_mul(T6, ji->dAdq, ji->invA); m6_mul(tmp1, T6, ji->Adot);
            m6_mul(T6, ji->Adot, ji->invA); m6_mul(tmp2, T6, ji->dAdq);
            m6_mul(T6, Aleft, ji->dAdq); m6_mul(dAdq_BiBp, T6, Aright);
            for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) U[a][b] = ji->dAdotdq[a][b] - tmp1[a][b] - tmp2[a][b];
            m6_mul(T6, Aleft, U); m6_mul(dAdotdq_BiBp, T6, Aright);
        }
        for (int a = ji->parent; a >= 0; a = s->nd[a].parent) {    /* jointA loop :520-529 / :582-606 */
            onode* ja = &s->nd[a];
            if (!ja->ndof) continue;                     /* idxrA empty */
            const int cA = ja->idxR;
            double JPA[6], JdPA[6], o1[6], o2[6], o3[6];
            for (int r = 0; r < 6; r++) { JPA[r] = JX(s, rP + r, cA); JdPA[r] = JDX(s, rP + r, cA); }
            blk_mul_col(o1, A_BiBp, JPA);
            blk_mul_col(o2, A_BiBp, JdPA);
            blk_mul_col(o3, Adot_BiBp, JPA);
            for (int r = 0; r < 6; r++) { JX(s, rI + r, cA) = o1[r]; JDX(s, rI + r, cA) = o2[r] + o3[r]; }
            if (!deriv) continue;
            if (ji->ndof) {                              /* :588-593 */
                const int k = ji->idxR;
                blk_mul_col(o1, dAdq_BiBp, JPA);
                blk_mul_col(o2, dAdq_BiBp, JdPA);
                blk_mul_col(o3, dAdotdq_BiBp, JPA);
                for (int r = 0; r < 6; r++) { T3(s->dJdq, s, rI + r, cA, k) = o1[r]; T3(s->dJdotdq, s, rI + r, cA, k) = o2[r] + o3[r]; }
            }
            for (int kk = ji->parent; kk >= 0; kk = s->nd[kk].parent) {   /* jointK loop :594-604 */
                onode* jk = &s->nd[kk];
                if (!jk->ndof) continue;
                const int k = jk->idxR;
                double dP[6], ddP[6];
                for (int r = 0; r < 6; r++) { dP[r] = T3(s->dJdq, s, rP + r, cA, k); ddP[r] = T3(s->dJdotdq, s, rP + r, cA, k); }
                blk_mul_col(o1, A_BiBp, dP);
                blk_mul_col(o2, A_BiBp, ddP);
                blk_mul_col(o3, Adot_BiBp, dP);
                for (int r = 0; r < 6; r++) { T3(s->dJdq, s, rI + r, cA, k) = o1[r]; T3(s->dJdotdq, s, rI + r, cA, k) = o2[r] + o3[r]; }
            }
        }
    }
}

void orc_jacobian(orc_scene* s, double* J, double* Jdot, double* dJdq, double* dJdotdq) {
    int deriv = (dJdq || dJdotdq);
    compute_jacobian(s, deriv);
    size_t a = (size_t)s->nm * s->nr;
    if (J) memcpy(J, s->J, a * sizeof(double));
    if (Jdot) memcpy(Jdot, s->Jdot, a * sizeof(double));
    if (dJdq) memcpy(dJdq, s->dJdq, a * s->nr * sizeof(double));
    if (dJdotdq) memcpy(dJdotdq, s->dJdotdq, a * s->nr * sizeof(double));
}

/* ----------------------------------- Body.computeMassGrav, Joint.computeForce */

/* Body.computeMassGrav (Body.m:83-135).  Blocks stored per body (block index = joint index). */
static void compute_mass_grav(orc_scene* s, int deriv) {
    for (int i = 0; i < s->n; i++) {
        onode* j = &s->nd[i];
        double* Mb = s->Mm + 36 * i; double* Kb = s->Km + 36 * i; double* Db = s->Dm + 36 * i;
        memset(Mb, 0, 36 * sizeof(double)); memset(Kb, 0, 36 * sizeof(double)); memset(Db, 0, 36 * sizeof(double));
        for (int a = 0; a < 6; a++) Mb[a * 6 + a] = j->I_i[a];
        m6 adm; se3_ad(adm, j->phi);
        /* adt = ad(phi)' ; fcor = adt*M_i*phi */
        double Mphi[6], fcor[6];
        for (int a = 0; a < 6; a++) Mphi[a] = j->I_i[a] * j->phi[a];
        for (int a = 0; a < 6; a++) { double t = 0; for (int k = 0; k < 6; k++) t += adm[k][a] * Mphi[k]; fcor[a] = t; }
        double mass = j->I_i[3];
        double grav_i[3];
        for (int a = 0; a < 3; a++) { double t = 0; for (int k = 0; k < 3; k++) t += j->E_wi[k][a] * s->grav[k]; grav_i[a] = t; }
        double fgrav[6] = { 0, 0, 0, mass * grav_i[0], mass * grav_i[1], mass * grav_i[2] };
        for (int a = 0; a < 6; a++) s->fm[j->idxM + a] = fcor[a] + fgrav[a];
        if (!deriv) continue;
        /* Km(rows(4:6),rows(1:3)) += brac(fgrav(4:6))      Body.m:119 */
        double fb[3][3]; se3_brac3(fb, fgrav + 3);
        for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) Kb[(3 + a) * 6 + b] += fb[a][b];
        /* Dm(rows,rows) += adt*M_i - [e1*Iw e2*Iw e3*Iw e1*mv e2*mv e3*mv ; e1*mv e2*mv e3*mv 0 0 0]   Body.m:121-129 */
        double Iw[3] = { j->I_i[0] * j->phi[0], j->I_i[1] * j->phi[1], j->I_i[2] * j->phi[2] };
        double mv[3] = { mass * j->phi[3], mass * j->phi[4], mass * j->phi[5] };
        for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) Db[a * 6 + b] += adm[b][a] * j->I_i[b];
        for (int c = 0; c < 3; c++) {
            double e[3] = { 0, 0, 0 }; e[c] = 1.0;
            double eb[3][3]; se3_brac3(eb, e);
            for (int a = 0; a < 3; a++) {
                double eIw = 0, emv = 0;
                for (int k = 0; k < 3; k++) { eIw += eb[a][k] * Iw[k]; emv += eb[a][k] * mv[k]; }
                Db[a * 6 + c] -= eIw;            /* top-left:  e_c*Iw */
                Db[a * 6 + 3 + c] -= emv;        /* top-right: e_c*mv */
                Db[(3 + a) * 6 + c] -= emv;      /* bottom-left: e_c*mv */
            }
        }
    }
}

/* ForceGroundCuboid.computeValues_ (ForceGroundCuboid.m:54-153): Geilinger-style penalty contact of the 8 cuboid corners with
 * the ground plane (frame E, Z up): normal spring-damper, static/dynamic friction; adds to fm and, if deriv, to the body's
 * 6x6 blocks of Km and Dm.  3x3 / 3x6 helpers are written out; G = Gamma(xl) = [brac(xl)', I] (se3.m:38-41). */
static void m33_mul(double C[3][3], const double A[3][3], const double B[3][3]) {
    double T[3][3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double t = 0; for (int k = 0; k < 3; k++) t += A[i][k] * B[k][j]; T[i][j] = t; }
    memcpy(C, T, sizeof(T));
}
static void m33_mulv(double y[3], const double A[3][3], const double x[3]) {
    double t[3];
    for (int i = 0; i < 3; i++) t[i] = A[i][0] * x[0] + A[i][1] * x[1] + A[i][2] * x[2];
    memcpy(y, t, sizeof(t));
}
/* Blk(6x6) += sign * G' * X(3x6) */
static void add_Gt_X(double* Blk, double sign, const double G[3][6], const double X[3][6]) {
    for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) {
        double t = 0; for (int k = 0; k < 3; k++) t += G[k][a] * X[k][b];
        Blk[a * 6 + b] += sign * t;
    }
}
static void compute_ground_contact(orc_scene* s, int deriv) {
    if (!s->contact) return;
    double eb[3][3][3];
    for (int c = 0; c < 3; c++) { double e[3] = { 0, 0, 0 }; e[c] = 1.0; se3_brac3(eb[c], e); }
    for (int fi = 0; fi < s->n + s->nxf; fi++) {       /* Force.computeValues walks the list of force objects (Force.m:26-56) */
        const int ib = fi < s->n ? fi : s->xf_body[fi - s->n];
        if (fi < s->n && !s->contact[ib]) continue;
        onode* j = &s->nd[ib];
        /* every force object holds its own E, kn, kt, mu, kd (ForceGroundCuboid.m:6-13, 56-57) */
        const struct ogf* gf = fi < s->n ? &s->gf[ib] : &s->xf[fi - s->n];
        const double kn = gf->kn, kt = gf->kt, mu = gf->mu, kd = gf->kd;
        double xg[3], ng[3], N[3][3], T[3][3];
        for (int a = 0; a < 3; a++) { xg[a] = gf->E[a][3]; ng[a] = gf->E[a][2]; }
        for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) { N[a][b] = ng[a] * ng[b]; T[a][b] = (a == b) - N[a][b]; }
        double* fm = s->fm + j->idxM; double* Kb = s->Km + 36 * ib; double* Db = s->Dm + 36 * ib;
        double R[3][3], Rt[3][3], p[3], RNR[3][3], RtN[3][3], B[3][3], RtT[3][3], pxgtmp[3][3], tv[3];
        for (int a = 0; a < 3; a++) { for (int b = 0; b < 3; b++) { R[a][b] = j->E_wi[a][b]; Rt[b][a] = R[a][b]; } p[a] = j->E_wi[a][3]; }
        m33_mul(RtN, Rt, N); m33_mul(RNR, RtN, R);             /* RNR = R'*N*R */
        m33_mul(RtT, Rt, T); m33_mul(B, RtT, R);               /* B = R'*T*R   */
        double pm[3] = { p[0] - xg[0], p[1] - xg[1], p[2] - xg[2] };
        m33_mulv(tv, RtN, pm); se3_brac3(pxgtmp, tv);          /* pxgtmp = brac(R'*N*(p - xg)) */
        const double* phi = j->phi;
        for (int ic = 0; ic < 8; ic++) {
            double xli[3] = { ((ic & 4) ? 0.5 : -0.5) * s->sides[3 * ib], ((ic & 2) ? 0.5 : -0.5) * s->sides[3 * ib + 1],
                              ((ic & 1) ? 0.5 : -0.5) * s->sides[3 * ib + 2] };
            double xwi[3]; m33_mulv(xwi, R, xli); for (int a = 0; a < 3; a++) xwi[a] += p[a];
            double d = 0; for (int a = 0; a < 3; a++) d += ng[a] * (xwi[a] - xg[a]);
            if (d > 0) continue;                                /* no collision */
            double xlb[3][3]; se3_brac3(xlb, xli);
            double G[3][6];
            for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) { G[a][b] = xlb[b][a]; G[a][3 + b] = (a == b); }
            double Gphi[3]; for (int a = 0; a < 3; a++) { double t = 0; for (int b = 0; b < 6; b++) t += G[a][b] * phi[b]; Gphi[a] = t; }
            double vwi[3]; m33_mulv(vwi, R, Gphi);
            /* fc = -kn*ng*d - kd*N*vwi ; fm += G'*R'*fc */
            double Nv[3]; m33_mulv(Nv, N, vwi);
            double fc[3], Rtf[3];
            for (int a = 0; a < 3; a++) fc[a] = -kn * ng[a] * d - kd * Nv[a];
            m33_mulv(Rtf, Rt, fc);
            for (int a = 0; a < 6; a++) { double t = 0; for (int k = 0; k < 3; k++) t += G[k][a] * Rtf[k]; fm[a] += t; }
            if (deriv) {
                double RNRxl[3], RNRGphi[3], Gpb[3][3], X[3][6], t33[3][3];
                m33_mulv(RNRxl, RNR, xli); m33_mulv(RNRGphi, RNR, Gphi); se3_brac3(Gpb, Gphi);
                /* tmp1 = -[e1b*RNRxl,e2b*RNRxl,e3b*RNRxl] - RNR*xlbrac + pxgtmp ; Km -= kn*G'*[tmp1 RNR] */
                m33_mul(t33, RNR, xlb);
                for (int a = 0; a < 3; a++) for (int c = 0; c < 3; c++) {
                    double ec = eb[c][a][0] * RNRxl[0] + eb[c][a][1] * RNRxl[1] + eb[c][a][2] * RNRxl[2];
                    X[a][c] = -ec - t33[a][c] + pxgtmp[a][c];
                    X[a][3 + c] = RNR[a][c];
                }
                add_Gt_X(Kb, -kn, G, X);
                /* tmp2 = -[e1b*RNRGphi,...] - RNR*Gphibrac ; Km -= kd*G'*[tmp2 0] */
                m33_mul(t33, RNR, Gpb);
                for (int a = 0; a < 3; a++) for (int c = 0; c < 3; c++) {
                    double ec = eb[c][a][0] * RNRGphi[0] + eb[c][a][1] * RNRGphi[1] + eb[c][a][2] * RNRGphi[2];
                    X[a][c] = -ec - t33[a][c];
                    X[a][3 + c] = 0.0;
                }
                add_Gt_X(Kb, -kd, G, X);
                /* Dm -= kd*G'*RNR*G */
                for (int a = 0; a < 3; a++) for (int c = 0; c < 6; c++) { double t = 0; for (int k = 0; k < 3; k++) t += RNR[a][k] * G[k][c]; X[a][c] = t; }
                add_Gt_X(Db, -kd, G, X);
            }
            if (mu == 0) continue;
            /* friction: a = T*xwdot, xwdot = R*G*phi */
            double av[3]; m33_mulv(av, T, vwi);
            double anorm = sqrt(av[0] * av[0] + av[1] * av[1] + av[2] * av[2]);
            if (mu * fabs(kn * d) > kt * anorm) {      /* static friction: fs = -kt*a */
                double fs[3] = { -kt * av[0], -kt * av[1], -kt * av[2] };
                m33_mulv(Rtf, Rt, fs);
                for (int a = 0; a < 6; a++) { double t = 0; for (int k = 0; k < 3; k++) t += G[k][a] * Rtf[k]; fm[a] += t; }
                if (deriv) {
                    double X[3][6];
                    /* D = -kt*G'*R'*T*R*G = -kt*G'*B*G */
                    for (int a = 0; a < 3; a++) for (int c = 0; c < 6; c++) { double t = 0; for (int k = 0; k < 3; k++) t += B[a][k] * G[k][c]; X[a][c] = t; }
                    add_Gt_X(Db, -kt, G, X);
                    /* K = -kt*G'*[(B*e1b-e1b*B)*Gphi, (B*e2b-e2b*B)*Gphi, (B*e3b-e3b*B)*Gphi, Z] */
                    for (int c = 0; c < 3; c++) {
                        double Be[3][3], eB[3][3], col[3];
                        m33_mul(Be, B, eb[c]); m33_mul(eB, eb[c], B);
                        for (int a = 0; a < 3; a++) for (int k = 0; k < 3; k++) Be[a][k] -= eB[a][k];
                        m33_mulv(col, Be, Gphi);
                        for (int a = 0; a < 3; a++) { X[a][c] = col[a]; X[a][3 + c] = 0.0; }
                    }
                    add_Gt_X(Kb, -kt, G, X);
                }
            } else {                                            /* dynamic friction: fd = -mu*kn*d*t */
                const double mukn = mu * kn;
                double tt[3] = { av[0] / anorm, av[1] / anorm, av[2] / anorm };
                double fd[3] = { -mukn * d * tt[0], -mukn * d * tt[1], -mukn * d * tt[2] };
                m33_mulv(Rtf, Rt, fd);
                for (int a = 0; a < 6; a++) { double t = 0; for (int k = 0; k < 3; k++) t += G[k][a] * Rtf[k]; fm[a] += t; }
                if (deriv) {
                    /* A = (a'*a*I - a*a')/norm(a)^3 */
                    double A[3][3], aa = av[0] * av[0] + av[1] * av[1] + av[2] * av[2], n3 = anorm * anorm * anorm;
                    for (int a = 0; a < 3; a++) for (int c = 0; c < 3; c++) A[a][c] = (aa * (a == c) - av[a] * av[c]) / n3;
                    double AT[3][3], RtAT[3][3], RtATR[3][3], X[3][6];
                    m33_mul(AT, A, T); m33_mul(RtAT, Rt, AT); m33_mul(RtATR, RtAT, R);
                    /* D = -mukn*G'*R'*d*A*T*R*G */
                    for (int a = 0; a < 3; a++) for (int c = 0; c < 6; c++) { double t = 0; for (int k = 0; k < 3; k++) t += RtATR[a][k] * G[k][c]; X[a][c] = d * t; }
                    add_Gt_X(Db, -mukn, G, X);
                    /* K1 = -d*[e1b*R't, e2b*R't, e3b*R't, Z]; K2 = R'*t*ng'*R*G; K3 = -d*R'*A*T*R*[brac(G*phi), Z]; K = -mukn*G'*(K1+K2+K3) */
                    double Rtt[3], ngR[3], Gpb[3][3], K3[3][3];
                    m33_mulv(Rtt, Rt, tt);
                    for (int c = 0; c < 3; c++) ngR[c] = ng[0] * R[0][c] + ng[1] * R[1][c] + ng[2] * R[2][c];
                    se3_brac3(Gpb, Gphi); m33_mul(K3, RtATR, Gpb);
                    double ngRG[6];
                    for (int c = 0; c < 6; c++) ngRG[c] = ngR[0] * G[0][c] + ngR[1] * G[1][c] + ngR[2] * G[2][c];
                    for (int a = 0; a < 3; a++) {
                        for (int c = 0; c < 3; c++) {
                            double ec = eb[c][a][0] * Rtt[0] + eb[c][a][1] * Rtt[1] + eb[c][a][2] * Rtt[2];
                            X[a][c] = -d * ec + Rtt[a] * ngRG[c] - d * K3[a][c];
                            X[a][3 + c] = Rtt[a] * ngRG[3 + c];
                        }
                    }
                    add_Gt_X(Kb, -mukn, G, X);
                }
            }
        }
    }
}
/* ForceGroundCuboid.computeEnergy_ (ForceGroundCuboid.m:156-183) */
static double ground_contact_energy(const orc_scene* s) {
    if (!s->contact) return 0.0;
    double v = 0;
    for (int fi = 0; fi < s->n + s->nxf; fi++) {
        const int ib = fi < s->n ? fi : s->xf_body[fi - s->n];
        if (fi < s->n && !s->contact[ib]) continue;
        const struct ogf* gf = fi < s->n ? &s->gf[ib] : &s->xf[fi - s->n];
        const onode* j = &s->nd[ib];
        for (int ic = 0; ic < 8; ic++) {
            double xli[3] = { ((ic & 4) ? 0.5 : -0.5) * s->sides[3 * ib], ((ic & 2) ? 0.5 : -0.5) * s->sides[3 * ib + 1],
                              ((ic & 1) ? 0.5 : -0.5) * s->sides[3 * ib + 2] };
            double d = 0;
            for (int a = 0; a < 3; a++) {
                double xw = j->E_wi[a][0] * xli[0] + j->E_wi[a][1] * xli[1] + j->E_wi[a][2] * xli[2] + j->E_wi[a][3];
                d += gf->E[a][2] * (xw - gf->E[a][3]);
            }
            if (d > 0) continue;
            v += 0.5 * gf->kn * (d * d);
        }
    }
    return v;
}
/* Attach ForceGroundCuboid to the flagged bodies (scenesRedMax.m:307-311 style: setTransform, setStiffness, setDamping,
 * setFriction).  E16 column-major. */
void orc_set_ground_contact(orc_scene* s, const int* flags, const double* sides, const double* E16, double kn, double kt, double mu, double kd) {
    orc_set_ground_contact_body(s, flags, sides, E16, 0, &kn, &kt, &mu, &kd, 0);
}
/* One force object per flagged body, each with its own frame and constants (ForceGroundCuboid.m:28-47).  E16: [n][16] column-major
 * when E_per_body, else one frame for all; kn..kd: [n] when k_per_body, else one value each. */
void orc_set_ground_contact_body(orc_scene* s, const int* flags, const double* sides, const double* E16, int E_per_body,
                                 const double* kn, const double* kt, const double* mu, const double* kd, int k_per_body) {
    free(s->contact); free(s->sides); free(s->gf);
    free(s->xf_body); free(s->xf); s->xf_body = NULL; s->xf = NULL; s->nxf = 0;
    s->contact = (int*)malloc(sizeof(int) * (size_t)s->n);
    s->sides = (double*)malloc(sizeof(double) * 3 * (size_t)s->n);
    s->gf = (struct ogf*)calloc((size_t)s->n, sizeof(struct ogf));
    memcpy(s->contact, flags, sizeof(int) * (size_t)s->n);
    memcpy(s->sides, sides, sizeof(double) * 3 * (size_t)s->n);
    for (int i = 0; i < s->n; i++) {
        cm16_to_m4(s->gf[i].E, E16 + (E_per_body ? 16 * (size_t)i : 0));
        const int k = k_per_body ? i : 0;
        s->gf[i].kn = kn[k]; s->gf[i].kt = kt[k]; s->gf[i].mu = mu[k]; s->gf[i].kd = kd[k];
    }
    orc_reset(s);
}

/* One more ForceGroundCuboid object on a body that orc_set_ground_contact[_body] has flagged (its sides are taken from there): the
 * reference appends force objects to a list (scenesRedMax.m:303-309 `scene.forces{end+1} = ...`), any number per body. */
int orc_add_ground_contact(orc_scene* s, int body, const double* E16, double kn, double kt, double mu, double kd) {
    if (!s->contact || body < 0 || body >= s->n || !s->contact[body]) return -1;
    s->xf_body = (int*)realloc(s->xf_body, sizeof(int) * (size_t)(s->nxf + 1));
    s->xf = (struct ogf*)realloc(s->xf, sizeof(struct ogf) * (size_t)(s->nxf + 1));
    s->xf_body[s->nxf] = body;
    cm16_to_m4(s->xf[s->nxf].E, E16);
    s->xf[s->nxf].kn = kn; s->xf[s->nxf].kt = kt; s->xf[s->nxf].mu = mu; s->xf[s->nxf].kd = kd;
    s->nxf++;
    orc_reset(s);
    return 0;
}

/* Joint.computeForce (Joint.m:437-487).  Kr, Dr are diagonal for 1-DOF joints. */
static void compute_joint_force(orc_scene* s) {
    for (int i = 0; i < s->nr; i++) { s->fr[i] = 0; s->Kr[i] = 0; s->Dr[i] = 0; }
    for (int i = 0; i < s->n; i++) {
        onode* j = &s->nd[i];
        if (!j->ndof) continue;
        int r = j->idxR;
        s->fr[r] += j->tau + j->stiffness * (j->qRest - j->q) - j->damping * j->qdot;
        s->Kr[r] -= j->stiffness;
        s->Dr[r] -= j->damping;
        double hitL = (j->q < j->qLimL) ? 1.0 : 0.0;
        double hitU = (j->q > j->qLimU) ? 1.0 : 0.0;
        s->fr[r] += hitL * (j->qLimK * (j->qLimL - j->q) - j->qLimD * j->qdot);
        s->fr[r] += hitU * (j->qLimK * (j->qLimU - j->q) - j->qLimD * j->qdot);
        s->Kr[r] -= (hitL * hitL) * j->qLimK;
        s->Kr[r] -= (hitU * hitU) * j->qLimK;
        s->Dr[r] -= (hitL * hitL) * j->qLimD;
        s->Dr[r] -= (hitU * hitU) * j->qLimD;
    }
}

/* ------------------------------------------------ computeValues (driver) */

/* y(nm) = Blk * x(nm) with Blk block-diagonal, blocks indexed by joint (rows idxM..idxM+5) */
static void blkdiag_mulv(const orc_scene* s, const double* Blk, const double* x, double* y) {
    for (int i = 0; i < s->n; i++) {
        const double* B = Blk + 36 * i; int r0 = s->nd[i].idxM;
        for (int a = 0; a < 6; a++) { double t = 0; for (int k = 0; k < 6; k++) t += B[a * 6 + k] * x[r0 + k]; y[r0 + a] = t; }
    }
}
/* y(nr) = A' * x, A nm x nr col-major */
static void matT_mulv(const orc_scene* s, const double* A, const double* x, double* y) {
    for (int c = 0; c < s->nr; c++) {
        const double* col = A + (size_t)c * s->nm; double t = 0;
        for (int r = 0; r < s->nm; r++) { double v = col[r]; if (v != 0.0) t += v * x[r]; }
        y[c] = t;
    }
}
/* y(nm) = A * x, A nm x nr col-major */
static void mat_mulv(const orc_scene* s, const double* A, const double* x, double* y) {
    for (int r = 0; r < s->nm; r++) y[r] = 0;
    for (int c = 0; c < s->nr; c++) {
        const double* col = A + (size_t)c * s->nm; double xc = x[c];
        if (xc == 0.0) continue;
        for (int r = 0; r < s->nm; r++) y[r] += col[r] * xc;
    }
}
/* C(nr x nr, col-major) = A' * (Blk * B), A,B nm x nr col-major */
static void AtBlkB(const orc_scene* s, const double* A, const double* Blk, const double* B, double* C, double* tmpcol) {
    for (int c = 0; c < s->nr; c++) {
        blkdiag_mulv(s, Blk, B + (size_t)c * s->nm, tmpcol);
        matT_mulv(s, A, tmpcol, C + (size_t)c * s->nr);
    }
}

/* computeValues (driverRedMaxBDF1.m:190-243) */
void orc_compute_values(orc_scene* s, double* M, double* f, double* dMdq, double* K, double* D) {
    const int nr = s->nr, nm = s->nm;
    const int deriv = (dMdq || K || D);
    double* qdot = (double*)calloc((size_t)nr + 1, sizeof(double));
    double* t_nm = (double*)calloc((size_t)nm, sizeof(double));
    double* t_nm2 = (double*)calloc((size_t)nm, sizeof(double));
    double* t_nr = (double*)calloc((size_t)nr + 1, sizeof(double));
    orc_get_state(s, NULL, qdot);
    compute_jacobian(s, deriv);
    compute_mass_grav(s, deriv);
    compute_ground_contact(s, deriv);      /* froot.computeValues (driverRedMaxBDF1.m:203,209) */
    compute_joint_force(s);
    /* M = J'*Mm*J   (:212) */
    if (M) AtBlkB(s, s->J, s->Mm, s->J, M, t_nm);
    /* fqvv = -J'*Mm*Jdot*qdot ; f = fr + J'*fm + fqvv   (:215-216) */
    double* MmJdotqdot = (double*)calloc((size_t)nm, sizeof(double));
    mat_mulv(s, s->Jdot, qdot, t_nm);
    blkdiag_mulv(s, s->Mm, t_nm, MmJdotqdot);
    if (f) {
        matT_mulv(s, s->J, s->fm, t_nr);
        double* t2 = (double*)calloc((size_t)nr + 1, sizeof(double));
        matT_mulv(s, s->J, MmJdotqdot, t2);
        for (int i = 0; i < nr; i++) f[i] = s->fr[i] + t_nr[i] + (-t2[i]);
        free(t2);
    }
    if (deriv) {
        double* tmp = (double*)calloc((size_t)nr * nr + 1, sizeof(double));
        double* Kqvv = (double*)calloc((size_t)nr * nr + 1, sizeof(double));
        double* Dqvv = (double*)calloc((size_t)nr * nr + 1, sizeof(double));
        /* dMdq(:,:,i) = tmp' + tmp, tmp = J'*Mm*dJdq(:,:,i)   (:220-224) */
        if (dMdq) for (int i = 0; i < nr; i++) {
            AtBlkB(s, s->J, s->Mm, s->dJdq + (size_t)i * nm * nr, tmp, t_nm);
            double* Di = dMdq + (size_t)i * nr * nr;
            for (int a = 0; a < nr; a++) for (int b = 0; b < nr; b++) Di[(size_t)b * nr + a] = tmp[(size_t)a * nr + b] + tmp[(size_t)b * nr + a];
        }
        /* Dqvv = -J'*Mm*Jdot  (:227) */
        AtBlkB(s, s->J, s->Mm, s->Jdot, Dqvv, t_nm);
        for (int i = 0; i < nr * nr; i++) Dqvv[i] = -Dqvv[i];
        for (int i = 0; i < nr; i++) {                 /* :229-234 */
            const double* dJdqi = s->dJdq + (size_t)i * nm * nr;
            const double* dJdotdqi = s->dJdotdq + (size_t)i * nm * nr;
            /* Kqvv(:,i) = -dJdqi'*MmJdotqdot - J'*Mm*dJdotdqi*qdot */
            matT_mulv(s, dJdqi, MmJdotqdot, t_nr);
            mat_mulv(s, dJdotdqi, qdot, t_nm); blkdiag_mulv(s, s->Mm, t_nm, t_nm2);
            double* t2 = (double*)calloc((size_t)nr + 1, sizeof(double));
            matT_mulv(s, s->J, t_nm2, t2);
            for (int a = 0; a < nr; a++) Kqvv[(size_t)i * nr + a] = -t_nr[a] - t2[a];
            /* Dqvv(:,i) = Dqvv(:,i) - J'*Mm*dJdqi*qdot */
            mat_mulv(s, dJdqi, qdot, t_nm); blkdiag_mulv(s, s->Mm, t_nm, t_nm2);
            matT_mulv(s, s->J, t_nm2, t2);
            for (int a = 0; a < nr; a++) Dqvv[(size_t)i * nr + a] -= t2[a];
            free(t2);
        }
        /* K = Kr + J'*Km*J + Kqvv ; D = Dr + J'*Dm*J + Dqvv  (:236-237) */
        if (K) {
            AtBlkB(s, s->J, s->Km, s->J, K, t_nm);
            for (int i = 0; i < nr * nr; i++) K[i] += Kqvv[i];
            for (int i = 0; i < nr; i++) K[(size_t)i * nr + i] += s->Kr[i];
            for (int i = 0; i < nr; i++) {             /* :238-241 */
                const double* dJdqi = s->dJdq + (size_t)i * nm * nr;
                matT_mulv(s, dJdqi, s->fm, t_nr);
                mat_mulv(s, dJdqi, qdot, t_nm); blkdiag_mulv(s, s->Dm, t_nm, t_nm2);
                double* t2 = (double*)calloc((size_t)nr + 1, sizeof(double));
                matT_mulv(s, s->J, t_nm2, t2);
                for (int a = 0; a < nr; a++) K[(size_t)i * nr + a] += t_nr[a] + t2[a];
                free(t2);
            }
        }
        if (D) {
            AtBlkB(s, s->J, s->Dm, s->J, D, t_nm);
            for (int i = 0; i < nr * nr; i++) D[i] += Dqvv[i];
            for (int i = 0; i < nr; i++) D[(size_t)i * nr + i] += s->Dr[i];
        }
        free(tmp); free(Kqvv); free(Dqvv);
    }
    free(MmJdotqdot); free(qdot); free(t_nm); free(t_nm2); free(t_nr);
}

/* evalBDF1 (driverRedMaxBDF1.m:160-187) and its BDF2/SDIRK2 siblings (driverRedMaxBDF2.m:194-293)
 * in the common form  qdot=(q-qA)/eta, dqtmp=q-qB, g=M*dqtmp-eta^2 f, H=M-eta*D-eta^2*K+dMdq*dqtmp. */
void orc_eval_residual(orc_scene* s, const double* q, const double* qA, const double* qB, double eta, double* g, double* H) {
    const int nr = s->nr;
    size_t n2 = (size_t)nr * nr + 1;
    double* qdot = (double*)calloc((size_t)nr + 1, sizeof(double));
    double* dq = (double*)calloc((size_t)nr + 1, sizeof(double));
    double* M = (double*)calloc(n2, sizeof(double));
    double* f = (double*)calloc((size_t)nr + 1, sizeof(double));
    for (int i = 0; i < nr; i++) { dq[i] = q[i] - qB[i]; qdot[i] = (q[i] - qA[i]) / eta; }
    set_q(s, q, qdot);
    scene_update(s);
    const double e2 = eta * eta;
    if (!H) {
        orc_compute_values(s, M, f, NULL, NULL, NULL);
        for (int a = 0; a < nr; a++) { double t = 0; for (int b = 0; b < nr; b++) t += M[(size_t)b * nr + a] * dq[b]; g[a] = t - e2 * f[a]; }
    } else {
        double* K = (double*)calloc(n2, sizeof(double));
        double* D = (double*)calloc(n2, sizeof(double));
        double* dMdq = (double*)calloc((size_t)nr * nr * nr + 1, sizeof(double));
        orc_compute_values(s, M, f, dMdq, K, D);
        for (int a = 0; a < nr; a++) { double t = 0; for (int b = 0; b < nr; b++) t += M[(size_t)b * nr + a] * dq[b]; g[a] = t - e2 * f[a]; }
        for (size_t i = 0; i < (size_t)nr * nr; i++) H[i] = M[i] - eta * D[i] - e2 * K[i];
        for (int i = 0; i < nr; i++) {
            const double* Di = dMdq + (size_t)i * nr * nr;
            for (int a = 0; a < nr; a++) { double t = 0; for (int b = 0; b < nr; b++) t += Di[(size_t)b * nr + a] * dq[b]; H[(size_t)i * nr + a] += t; }
        }
        free(K); free(D); free(dMdq);
    }
    free(qdot); free(dq); free(M); free(f);
}

/* ------------------------------------------------------------- newton */

/* dx = -H\g : dense LU with partial pivoting (MATLAB mldivide -> LAPACK dgetrf/dgetrs;
 * call site driverRedMaxBDF1.m:117).  H is col-major and is destroyed. Returns 0 if singular. */
static int lu_solve_neg(int n, double* H, const double* g, double* dx) {
    int* piv = (int*)malloc(sizeof(int) * (size_t)(n + 1));
    double* b = (double*)malloc(sizeof(double) * (size_t)(n + 1));
    for (int i = 0; i < n; i++) b[i] = -g[i];
#define Hc(r, c) H[(size_t)(c) * n + (r)]
    int ok = 1;
    for (int k = 0; k < n; k++) {
        int p = k; double mx = fabs(Hc(k, k));
        for (int r = k + 1; r < n; r++) { double v = fabs(Hc(r, k)); if (v > mx) { mx = v; p = r; } }
        piv[k] = p;
        if (mx == 0.0) { ok = 0; continue; }
        if (p != k) {
            for (int c = 0; c < n; c++) { double t = Hc(k, c); Hc(k, c) = Hc(p, c); Hc(p, c) = t; }
            double t = b[k]; b[k] = b[p]; b[p] = t;
        }
        double inv = 1.0 / Hc(k, k);
        for (int r = k + 1; r < n; r++) Hc(r, k) *= inv;
        for (int c = k + 1; c < n; c++) {
            double hkc = Hc(k, c);
            if (hkc != 0.0) for (int r = k + 1; r < n; r++) Hc(r, c) -= Hc(r, k) * hkc;
        }
    }
    for (int k = 0; k < n; k++) for (int r = k + 1; r < n; r++) b[r] -= Hc(r, k) * b[k];    /* L y = b */
    for (int k = n - 1; k >= 0; k--) { b[k] /= Hc(k, k); for (int r = 0; r < k; r++) b[r] -= Hc(r, k) * b[k]; }
#undef Hc
    for (int i = 0; i < n; i++) dx[i] = b[i];
    free(piv); free(b);
    return ok;
}

static double vnorm(int n, const double* x) { double t = 0; for (int i = 0; i < n; i++) t += x[i] * x[i]; return sqrt(t); }

/* Newton constants: the reference hard-codes tol=1e-9, dxMax=1e3, iterMax=10*nr, iterLsMax=20
 * (driverRedMaxBDF1.m:95-98).  orc_set_newton lets tests/bench run the SAME algorithm with the SAME
 * constants the HIP library was given (rmx_opts). */
static double g_tol = 1e-9, g_dxMax = 1e3;
static int g_iterMaxPerDof = 10, g_iterLsMax = 20;
void orc_set_newton(double tol, double dxMax, int iterMaxPerDof, int iterLsMax) {
    g_tol = tol; g_dxMax = dxMax; g_iterMaxPerDof = iterMaxPerDof; g_iterLsMax = iterLsMax;
}
/* NOT in the reference: the library's opt-in straggler policy rmx_opts.ls_fail_limit restated, so that the option has a checker
 * too.  0 (default) = the reference's newton().  N > 0: the loop ends, "not converged", at the N-th line search of the call that
 * ran out its iterLsMax trials without a decrease of f. */
static int g_lsFailLimit = 0;
void orc_set_ls_fail_limit(int n) { g_lsFailLimit = n > 0 ? n : 0; }

/* Diagnostic, changes nothing newton() computes: per Newton iteration three doubles {|g| the iteration starts from, |g| its line
 * search ends with, trials the line search took} into a caller's buffer (single-threaded use only: Oracle.step_bdf1, not the OpenMP
 * batch).  tests/test_gpu_reference_tol.py logs them at the steps where the GPU's iteration count differs. */
static double* g_trace = NULL;
static int g_trace_cap = 0, g_trace_n = 0;
void orc_set_trace(double* buf, int cap_iters) { g_trace = buf; g_trace_cap = buf ? cap_iters : 0; g_trace_n = 0; }
int orc_trace_count(void) { return g_trace_n; }

/* newton (driverRedMaxBDF1.m:94-157): x updated in place */
static void newton(orc_scene* s, double* x, const double* qA, const double* qB, double eta, orc_stats* st) {
    const int nr = s->nr;
    const double tol = g_tol, dxMax = g_dxMax;
    const int iterMax = g_iterMaxPerDof * nr, iterLsMax = g_iterLsMax;
    double* g = (double*)calloc((size_t)nr + 1, sizeof(double));
    double* H = (double*)calloc((size_t)nr * nr + 1, sizeof(double));
    double* dx = (double*)calloc((size_t)nr + 1, sizeof(double));
    double* x0 = (double*)calloc((size_t)nr + 1, sizeof(double));
    int iter = 1, lsfail = 0;
    while (1) {
        orc_eval_residual(s, x, qA, qB, eta, g, H);
        if (st) { st->hessian_evals++; st->newton_iters++; }
        lu_solve_neg(nr, H, g, dx);
        if (vnorm(nr, dx) > dxMax) { if (st) st->diverged++; break; }
        double alpha = 1.0;
        double f0 = 0; for (int i = 0; i < nr; i++) f0 += g[i] * g[i]; f0 *= 0.5;
        memcpy(x0, x, sizeof(double) * (size_t)nr);
        int iterLs = 1, decreased = 0;
        while (1) {
            for (int i = 0; i < nr; i++) x[i] = x0[i] + alpha * dx[i];
            orc_eval_residual(s, x, qA, qB, eta, g, NULL);
            if (st) st->residual_evals++;
            double f = 0; for (int i = 0; i < nr; i++) f += g[i] * g[i]; f *= 0.5;
            if (f < f0) { decreased = 1; break; }
            if (iterLs >= iterLsMax) break;
            alpha = 0.5 * alpha;
            iterLs++;
        }
        if (st) st->ls_halvings += iterLs - 1;
        if (g_trace && g_trace_n < g_trace_cap) {
            g_trace[3 * g_trace_n] = sqrt(2.0 * f0);
            g_trace[3 * g_trace_n + 1] = vnorm(nr, g);
            g_trace[3 * g_trace_n + 2] = (double)iterLs;
            g_trace_n++;
        }
        if (vnorm(nr, g) < tol) break;
        if (iter >= iterMax) { if (st) { st->not_converged++; if (vnorm(nr, g) > st->worst_exit_g) st->worst_exit_g = vnorm(nr, g); } break; }
        lsfail += decreased ? 0 : 1;
        if (g_lsFailLimit > 0 && lsfail >= g_lsFailLimit) { if (st) st->not_converged++; break; }     /* orc_set_ls_fail_limit */
        iter++;
    }
    free(g); free(H); free(dx); free(x0);
}


/* ------------------------------------------------------------ JointSpherical: Euler-angle charts and reparam
 *
 * JointSpherical (matlab-diff/+redmax/JointSpherical.m) parameterises the rotation by one of 12 Euler charts, R = R_a1(q1)
 * R_a2(q2) R_a3(q3) (codegen :247-262: XYX = X1*Y2*X3, ..., XYZ = X1*Y2*Z3, ...), i.e. by a chain of three revolute joints about
 * the chart's axes; S(1:3,1:3) = T with T(:,i) = vee(R' dR/dq_i) (:298-303) is that chain's body-frame Jacobian.  The oracle
 * holds such a joint as its three 1-DOF nodes (two massless links) and restates what is NOT a property of the chain:
 * reparam_ (:63-102), the chart switch after a step when |det T| <= 0.5, with getEulerInv (:181-208, XYXinv..ZYXinv :1809-1949).
 * JointFree3D (JointFree3D.m:16-34) = JointTranslational + JointSpherical.  Pinned by Hexpected of scenes 7 and 9
 * (scenesRedMax.m:206-207, 250-251); scene 7 under BDF2 switches charts. */
static const int CHART_AX[13][3] = { {0,0,0},
    {0,1,0}, {0,2,0}, {1,2,1}, {1,0,1}, {2,0,2}, {2,1,2},       /* XYX XZX YZY YXY ZXZ ZYZ   (JointSpherical.m:5-10)  */
    {0,1,2}, {0,2,1}, {1,2,0}, {1,0,2}, {2,0,1}, {2,1,0} };     /* XYZ XZY YZX YXZ ZXY ZYX   (:11-16)                 */

static void rot_elem(double R[3][3], int a, double q) {         /* X1/Y1/Z1 of codegen :247-255 */
    const double c = cos(q), s = sin(q);
    const int b = (a + 1) % 3, d = (a + 2) % 3;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[i][j] = (i == j) ? 1.0 : 0.0;
    R[b][b] = c; R[b][d] = -s; R[d][b] = s; R[d][d] = c;
}
static void mm3(double C[3][3], const double A[3][3], const double B[3][3]) {
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double t = 0; for (int k = 0; k < 3; k++) t += A[i][k] * B[k][j]; C[i][j] = t; }
}
static void euler_R(int chart, const double q[3], double R[3][3]) {
    double R1[3][3], R2[3][3], R3[3][3], T[3][3];
    rot_elem(R1, CHART_AX[chart][0], q[0]); rot_elem(R2, CHART_AX[chart][1], q[1]); rot_elem(R3, CHART_AX[chart][2], q[2]);
    mm3(T, R1, R2); mm3(R, T, R3);
}
/* T(:,1) = R3'R2' e_a1, T(:,2) = R3' e_a2, T(:,3) = e_a3 ; det T = +-cos q2 (Tait-Bryan) or +-sin q2 (proper Euler) */
static double euler_T(int chart, const double q[3], double T[3][3]) {
    double R2[3][3], R3[3][3];
    const int* ax = CHART_AX[chart];
    rot_elem(R2, ax[1], q[1]); rot_elem(R3, ax[2], q[2]);
    double e1[3] = {0, 0, 0}, e2[3] = {0, 0, 0}, t[3];
    e1[ax[0]] = 1.0; e2[ax[1]] = 1.0;
    for (int i = 0; i < 3; i++) { t[i] = 0; for (int k = 0; k < 3; k++) t[i] += R2[k][i] * e1[k]; }
    for (int i = 0; i < 3; i++) { T[i][0] = 0; for (int k = 0; k < 3; k++) T[i][0] += R3[k][i] * t[k]; }
    for (int i = 0; i < 3; i++) { T[i][1] = 0; for (int k = 0; k < 3; k++) T[i][1] += R3[k][i] * e2[k]; }
    for (int i = 0; i < 3; i++) T[i][2] = (i == ax[2]) ? 1.0 : 0.0;
    /* det T in closed form, as the reference's generated code has it (detS = simplify(det(S)), :304-306): +-sin q2 for the
     * proper-Euler charts, +-cos q2 for the Tait-Bryan ones (only |det T| is ever used, :66, :84).  The closed form matters:
     * XYX/XZX, YZY/YXY and ZXZ/ZYZ share q2 = acos(R_ii), so their |det T| tie EXACTLY and max() takes the first (:84);
     * a determinant evaluated from the matrix entries would break those ties by roundoff. */
    return (ax[2] == ax[0]) ? sin(q[1]) : cos(q[1]);
}
/* XYXinv .. ZYXinv (:1809-1949) in one rule.  With (i,j,k) = (a1, a2, remaining axis) and e = +1 if (i,j,k) is a cyclic
 * permutation of (x,y,z), -1 otherwise:
 *   proper Euler (a3 == a1): q2 = acos(R_ii),   q1 = atan2(R_ji, -e R_ki), q3 = atan2(R_ij,  e R_ik)
 *   Tait-Bryan   (a3 == k) : q2 = asin(e R_ik), q1 = atan2(-e R_jk, R_kk), q3 = atan2(-e R_ij, R_ii)
 * NaN at gimbal lock (the guarded entry not strictly inside (-1,1)), as the reference. */
static void euler_inv(int chart, const double R[3][3], double q[3]) {
    const int* ax = CHART_AX[chart];
    const int i = ax[0], j = ax[1], k = 3 - i - j;
    const double e = ((j - i + 3) % 3 == 1) ? 1.0 : -1.0;
    if (ax[2] == ax[0]) {
        const double r = R[i][i];
        if (-1.0 < r && r < 1.0) { q[0] = atan2(R[j][i], -e * R[k][i]); q[1] = acos(r); q[2] = atan2(R[i][j], e * R[i][k]); return; }
    } else {
        const double r = R[i][k];
        if (-1.0 < r && r < 1.0) { q[0] = atan2(-e * R[j][k], R[k][k]); q[1] = asin(e * r); q[2] = atan2(-e * R[i][j], R[i][i]); return; }
    }
    q[0] = q[1] = q[2] = NAN;
}
/* x = A\b for a 3x3 (MATLAB mldivide: LU with partial pivoting) */
static void solve3(const double Ain[3][3], const double b[3], double x[3]) {
    double A[3][4];
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) A[i][j] = Ain[i][j]; A[i][3] = b[i]; }
    for (int c = 0; c < 3; c++) {
        int p = c;
        for (int r = c + 1; r < 3; r++) if (fabs(A[r][c]) > fabs(A[p][c])) p = r;
        if (p != c) for (int j = 0; j < 4; j++) { double t = A[c][j]; A[c][j] = A[p][j]; A[p][j] = t; }
        for (int r = c + 1; r < 3; r++) { const double l = A[r][c] / A[c][c]; for (int j = c; j < 4; j++) A[r][j] -= l * A[c][j]; }
    }
    for (int r = 2; r >= 0; r--) { double t = A[r][3]; for (int j = r + 1; j < 3; j++) t -= A[r][j] * x[j]; x[r] = t / A[r][r]; }
}
static void sph_apply_chart(orc_scene* s, int g) {
    for (int k = 0; k < 3; k++) {
        onode* j = &s->nd[s->sph_first[g] + k];
        for (int a = 0; a < 3; a++) j->axis[a] = (a == CHART_AX[s->sph_chart[g]][k]) ? 1.0 : 0.0;
    }
}
/* Joint.reparam -> JointSpherical.reparam_ (:63-102), called after setQ at the end of every step (driverRedMaxBDF1.m:78,
 * driverRedMaxBDF2.m:112).  with_prev: the BDF2 drivers keep step k in q1/chart1 (setQ1/setAux1_), which reparam_ uses and
 * rewrites too.  The BDF1 driver never calls setQ1, so chart1 is empty there and the reference's reparam_ would fail at
 * getEuler(this.chart1,...) should a switch come up; here BDF1 chooses the chart from R alone (documented deviation; none of
 * the reference's BDF1 runs switches). */
static int reparam(orc_scene* s, int with_prev) {
    int switched = 0;
    for (int g = 0; g < s->nsph; g++) {
        onode* n0 = &s->nd[s->sph_first[g]];
        double q[3] = { n0[0].q, n0[1].q, n0[2].q }, qd[3] = { n0[0].qdot, n0[1].qdot, n0[2].qdot };
        double q1[3] = { n0[0].q1, n0[1].q1, n0[2].q1 }, qd1[3] = { n0[0].qdot1, n0[1].qdot1, n0[2].qdot1 };
        double Told[3][3], R[3][3], R1[3][3], Tt[3][3];
        const double detTold = euler_T(s->sph_chart[g], q, Told);
        if (fabs(detTold) > 0.5) continue;                                       /* :66-68 */
        euler_R(s->sph_chart[g], q, R);
        if (with_prev) euler_R(s->sph_chart1[g], q1, R1);                        /* :73 */
        int best = 1; double bestv = -1.0;
        for (int k = 1; k <= 12; k++) {                                          /* :75-83 */
            double qk[3], d0, d1 = INFINITY;
            euler_inv(k, R, qk); d0 = fabs(euler_T(k, qk, Tt)); if (isnan(d0)) d0 = 0.0;
            if (with_prev) { euler_inv(k, R1, qk); d1 = fabs(euler_T(k, qk, Tt)); if (isnan(d1)) d1 = 0.0; }
            const double v = d0 < d1 ? d0 : d1;
            if (v > bestv) { bestv = v; best = k; }                              /* max() keeps the first maximum */
        }
        double w[3], Tnew[3][3];
        for (int i = 0; i < 3; i++) w[i] = Told[i][0] * qd[0] + Told[i][1] * qd[1] + Told[i][2] * qd[2];
        const int chart_prev = s->sph_chart1[g], chart_old = s->sph_chart[g];
        s->sph_chart[g] = best;
        euler_inv(best, R, q);                                                   /* :87 */
        euler_T(best, q, Tnew);                                                  /* :89 */
        solve3(Tnew, w, qd);                                                     /* :91 */
        for (int k = 0; k < 3; k++) { n0[k].q = q[k]; n0[k].qdot = qd[k]; }
        if (with_prev) {                                                         /* :97-101 */
            euler_T(chart_prev, q1, Told);
            for (int i = 0; i < 3; i++) w[i] = Told[i][0] * qd1[0] + Told[i][1] * qd1[1] + Told[i][2] * qd1[2];
            euler_inv(best, R1, q1);
            euler_T(best, q1, Tnew);
            solve3(Tnew, w, qd1);
            for (int k = 0; k < 3; k++) { n0[k].q1 = q1[k]; n0[k].qdot1 = qd1[k]; }
        }
        s->sph_chart1[g] = best;
        sph_apply_chart(s, g);
        if (best != chart_old) switched++;
    }
    return switched;
}

/* groups: first[g] = index of the first of the three consecutive revolute nodes of spherical joint g (chart XYZ at start,
 * JointSpherical.m:33). */
int orc_set_spherical(orc_scene* s, int ngroups, const int* first) {
    for (int g = 0; g < ngroups; g++) {
        if (first[g] < 0 || first[g] + 2 >= s->n) return -1;
        for (int k = 0; k < 3; k++) {
            const onode* j = &s->nd[first[g] + k];
            if (j->type != ORC_JOINT_REVOLUTE || (k > 0 && j->parent != first[g] + k - 1)) return -1;
        }
    }
    free(s->sph_first); free(s->sph_chart); free(s->sph_chart1);
    s->nsph = ngroups;
    s->sph_first = (int*)calloc((size_t)ngroups + 1, sizeof(int));
    s->sph_chart = (int*)calloc((size_t)ngroups + 1, sizeof(int));
    s->sph_chart1 = (int*)calloc((size_t)ngroups + 1, sizeof(int));
    for (int g = 0; g < ngroups; g++) { s->sph_first[g] = first[g]; s->sph_chart[g] = s->sph_chart1[g] = 7; sph_apply_chart(s, g); }
    scene_update(s);
    return 0;
}
void orc_get_charts(const orc_scene* s, int* charts) { for (int g = 0; g < s->nsph; g++) charts[g] = s->sph_chart[g]; }
int orc_set_charts(orc_scene* s, const int* charts) {
    for (int g = 0; g < s->nsph; g++) if (charts[g] < 1 || charts[g] > 12) return -1;
    for (int g = 0; g < s->nsph; g++) { s->sph_chart[g] = s->sph_chart1[g] = charts[g]; sph_apply_chart(s, g); }
    scene_update(s);
    return 0;
}
/* the static helpers of JointSpherical for tests: R (row-major 3x3), T and det T of getEuler (:151-178), getEulerInv (:181-208) */
double orc_euler(int chart, const double* q, double* R9, double* T9) {
    double R[3][3], T[3][3];
    euler_R(chart, q, R);
    const double d = euler_T(chart, q, T);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { if (R9) R9[3 * i + j] = R[i][j]; if (T9) T9[3 * i + j] = T[i][j]; }
    return d;
}
void orc_euler_inv(int chart, const double* R9, double* q) {
    double R[3][3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[i][j] = R9[3 * i + j];
    euler_inv(chart, R, q);
}

/* simLoop (driverRedMaxBDF1.m:57-91) */
void orc_step_bdf1(orc_scene* s, double h, int nsteps, orc_stats* st, double* Hist_T, double* Hist_V) {
    const int nr = s->nr;
    double* q0 = (double*)calloc((size_t)nr + 1, sizeof(double));
    double* qd0 = (double*)calloc((size_t)nr + 1, sizeof(double));
    double* q1 = (double*)calloc((size_t)nr + 1, sizeof(double));
    double* qB = (double*)calloc((size_t)nr + 1, sizeof(double));
    double* qd1 = (double*)calloc((size_t)nr + 1, sizeof(double));
    if (st) memset(st, 0, sizeof(*st));
    for (int k = 0; k < nsteps; k++) {
        orc_get_state(s, q0, qd0);
        for (int i = 0; i < s->n; i++) { s->nd[i].q0 = s->nd[i].q; s->nd[i].qdot0 = s->nd[i].qdot; }  /* setQ0 */
        for (int i = 0; i < nr; i++) { q1[i] = q0[i] + h * qd0[i]; qB[i] = q0[i] + h * qd0[i]; }
        newton(s, q1, q0, qB, h, st);
        for (int i = 0; i < nr; i++) qd1[i] = (q1[i] - q0[i]) / h;
        set_q(s, q1, qd1);
        if (s->nsph && st) st->chart_switches += reparam(s, 0); else if (s->nsph) reparam(s, 0);   /* jroot.reparam() :78 */
        scene_update(s);
        if (Hist_T || Hist_V) { double T, V; orc_energy(s, &T, &V); if (Hist_T) Hist_T[k] = T; if (Hist_V) Hist_V[k] = V; }
    }
    free(q0); free(qd0); free(q1); free(qB); free(qd1);
}

/* simLoop (driverRedMaxBDF2.m:57-125): SDIRK2 start (evalSDIRK2a :194-225, evalSDIRK2b :228-260)
 * then BDF2 (evalBDF2 :263-293). Previous-step state is kept in nd[].q1/qdot1 (Joint.setQ1). */
void orc_step_bdf2(orc_scene* s, double h, int step0, int nsteps, orc_stats* st, double* Hist_T, double* Hist_V) {
    const int nr = s->nr;
    size_t sz = (size_t)nr + 1;
    double *q0 = calloc(sz, 8), *qd0 = calloc(sz, 8), *q1 = calloc(sz, 8), *qd1 = calloc(sz, 8);
    double *x = calloc(sz, 8), *qA = calloc(sz, 8), *qB = calloc(sz, 8), *xd = calloc(sz, 8);
    if (st) memset(st, 0, sizeof(*st));
    for (int k = step0; k < step0 + nsteps; k++) {
        if (k == 0) {
            const double a = (2.0 - sqrt(2.0)) / 2.0;
            orc_get_state(s, q0, qd0);
            /* SDIRK2a */
            for (int i = 0; i < nr; i++) { x[i] = q0[i] + a * h * qd0[i]; qA[i] = q0[i]; qB[i] = q0[i] + (a * h) * qd0[i]; }
            newton(s, x, qA, qB, a * h, st);
            double* qa = q1; double* qda = qd1;
            for (int i = 0; i < nr; i++) { qa[i] = x[i]; qda[i] = (x[i] - q0[i]) / (a * h); }
            /* SDIRK2b */
            for (int i = 0; i < nr; i++) {
                x[i] = qa[i] + (1 - a) * h * qda[i];
                qA[i] = q0[i] + (1 - a) * h * qda[i];
                qB[i] = q0[i] + (2 * a - 1) * h * qd0[i] + 2 * (1 - a) * h * qda[i];
            }
            newton(s, x, qA, qB, a * h, st);
            for (int i = 0; i < nr; i++) xd[i] = (x[i] - q0[i] - (1 - a) * h * qda[i]) / (a * h);
            set_q(s, x, xd);
            for (int i = 0; i < s->n; i++) if (s->nd[i].ndof) { s->nd[i].q1 = q0[s->nd[i].idxR]; s->nd[i].qdot1 = qd0[s->nd[i].idxR]; }  /* setQ1(q0,qdot0) */
        } else {
            for (int i = 0; i < s->n; i++) if (s->nd[i].ndof) { q0[s->nd[i].idxR] = s->nd[i].q1; qd0[s->nd[i].idxR] = s->nd[i].qdot1; }
            orc_get_state(s, q1, qd1);
            for (int i = 0; i < s->n; i++) { s->nd[i].q1 = s->nd[i].q; s->nd[i].qdot1 = s->nd[i].qdot; }
            for (int i = 0; i < nr; i++) {
                x[i] = q1[i] + h * qd1[i];
                qA[i] = (4.0 / 3.0) * q1[i] - (1.0 / 3.0) * q0[i];
                qB[i] = (4.0 / 3.0) * q1[i] - (1.0 / 3.0) * q0[i] + (8.0 / 9.0) * h * qd1[i] - (2.0 / 9.0) * h * qd0[i];
            }
            newton(s, x, qA, qB, (2.0 / 3.0) * h, st);
            for (int i = 0; i < nr; i++) xd[i] = (3.0 / (2.0 * h)) * (x[i] - (4.0 / 3.0) * q1[i] + (1.0 / 3.0) * q0[i]);
            set_q(s, x, xd);
        }
        if (s->nsph && st) st->chart_switches += reparam(s, 1); else if (s->nsph) reparam(s, 1);   /* jroot.reparam() :112 */
        scene_update(s);
        if (Hist_T || Hist_V) { double T, V; orc_energy(s, &T, &V); if (Hist_T) Hist_T[k - step0] = T; if (Hist_V) Hist_V[k - step0] = V; }
    }
    free(q0); free(qd0); free(q1); free(qd1); free(x); free(qA); free(qB); free(xd);
}

/* ---------------------------------------- matlab-simple euler (config 1) */

/* matlab-simple Joint.computeJacobian (matlab-simple/+redmax/Joint.m:250-305):
 * Jdot through world adjoints, Addot_ip = -Ad_iw*(Addot_wi*Ad_iw*Ad_wp - Addot_wp). */
static void compute_jacobian_simple(orc_scene* s) {
    const int nm = s->nm, nr = s->nr;
    memset(s->J, 0, sizeof(double) * (size_t)nm * nr);
    memset(s->Jdot, 0, sizeof(double) * (size_t)nm * nr);
    for (int i = 0; i < s->n; i++) {
        onode* ji = &s->nd[i];
        const int rI = ji->idxM;
        if (ji->ndof) {
            double col[6]; blk_mul_col(col, ji->A0_ij, ji->S);
            for (int r = 0; r < 6; r++) JX(s, rI + r, ji->idxR) = col[r];
        }
        if (ji->parent < 0) continue;
        onode* jp = &s->nd[ji->parent];
        const int rP = jp->idxM;
        m6 T1, T2, Addot_ip;
        m6_mul(T1, ji->Addot_wi, ji->Ad_iw); m6_mul(T2, T1, jp->Ad_wi);
        for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) T2[a][b] -= jp->Addot_wi[a][b];
        m6_mul(Addot_ip, ji->Ad_iw, T2);
        for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) Addot_ip[a][b] = -Addot_ip[a][b];
        for (int a = ji->parent; a >= 0; a = s->nd[a].parent) {
            onode* ja = &s->nd[a];
            if (!ja->ndof) continue;
            const int cA = ja->idxR;
            double JPA[6], JdPA[6], o1[6], o2[6], o3[6];
            for (int r = 0; r < 6; r++) { JPA[r] = JX(s, rP + r, cA); JdPA[r] = JDX(s, rP + r, cA); }
            blk_mul_col(o1, ji->Ad_ip, JPA);
            blk_mul_col(o2, ji->Ad_ip, JdPA);
            blk_mul_col(o3, Addot_ip, JPA);
            for (int r = 0; r < 6; r++) { JX(s, rI + r, cA) = o1[r]; JDX(s, rI + r, cA) = o2[r] + o3[r]; }
        }
    }
}

/* euler (matlab-simple/testRedMax.m:67-109). Scenes without body damping: Dm = 0. */
void orc_step_euler_simple(orc_scene* s, double h, int nsteps, double* Hist_T, double* Hist_V) {
    const int nr = s->nr, nm = s->nm;
    size_t sz = (size_t)nr + 1;
    double *q0 = calloc(sz, 8), *qd0 = calloc(sz, 8), *qd1 = calloc(sz, 8), *q1 = calloc(sz, 8);
    double *Mr = calloc((size_t)nr * nr + 1, 8), *Mt = calloc((size_t)nr * nr + 1, 8), *frt = calloc(sz, 8);
    double *t_nm = calloc((size_t)nm, 8), *t_nm2 = calloc((size_t)nm, 8), *t_nr = calloc(sz, 8);
    for (int k = 0; k < nsteps; k++) {
        compute_mass_grav(s, 0);
        compute_joint_force(s);   /* fr = tau - Kr*(q-qInit) - Dr*qdot ; Kr,Dr diagonal (Joint.m:212-247) */
        compute_jacobian_simple(s);
        orc_get_state(s, q0, qd0);
        AtBlkB(s, s->J, s->Mm, s->J, Mr, t_nm);
        for (int a = 0; a < nr; a++) for (int b = a + 1; b < nr; b++) {   /* Mr = 0.5*(Mr+Mr') */
            double v = 0.5 * (Mr[(size_t)b * nr + a] + Mr[(size_t)a * nr + b]);
            Mr[(size_t)b * nr + a] = v; Mr[(size_t)a * nr + b] = v;
        }
        /* frtilde = Mr*qdot0 + h*(J'*(fm - Mm*Jdot*qdot0) + fr) */
        mat_mulv(s, s->Jdot, qd0, t_nm); blkdiag_mulv(s, s->Mm, t_nm, t_nm2);
        for (int r = 0; r < nm; r++) t_nm2[r] = s->fm[r] - t_nm2[r];
        matT_mulv(s, s->J, t_nm2, t_nr);
        for (int a = 0; a < nr; a++) {
            double t = 0; for (int b = 0; b < nr; b++) t += Mr[(size_t)b * nr + a] * qd0[b];
            /* matlab-simple discards the damping FORCE ([~,Dr] = computeForceDamping, testRedMax.m:84):
             * fr_simple = tau - Kr*(q-qInit) = fr + damping*qdot = fr - Dr*qdot */
            frt[a] = t + h * (t_nr[a] + (s->fr[a] - s->Dr[a] * qd0[a]));
        }
        /* Mrtilde = Mr + J'*h*Dm*J + h*Dr - h*h*Kr ; matlab-simple sign convention: Dr=+damping, Kr=-stiffness;
         * compute_joint_force stores Dr=-damping, Kr=-stiffness (matlab-diff convention) => h*Dr_simple = -h*Dr. */
        memcpy(Mt, Mr, sizeof(double) * (size_t)nr * nr);
        for (int a = 0; a < nr; a++) Mt[(size_t)a * nr + a] += -h * s->Dr[a] - h * h * s->Kr[a];
        for (int a = 0; a < nr; a++) frt[a] = -frt[a];   /* lu_solve_neg solves Mt*x = -rhs */
        lu_solve_neg(nr, Mt, frt, qd1);
        for (int a = 0; a < nr; a++) q1[a] = q0[a] + h * qd1[a];
        set_q(s, q1, qd1);
        scene_update(s);
        if (Hist_T || Hist_V) { double T, V; orc_energy(s, &T, &V); if (Hist_T) Hist_T[k] = T; if (Hist_V) Hist_V[k] = V; }
    }
    free(q0); free(qd0); free(qd1); free(q1); free(Mr); free(Mt); free(frt); free(t_nm); free(t_nm2); free(t_nr);
}

/* ------------------------------------ adjoint BDF1 (config 4) */

/* [L,U,p] = lu(H,'vector') (driverRedMaxAdjointBDF1.m:127): in-place LU with partial pivoting and explicit row swaps,
 * H(p,:) = L*U, unit-lower L below the diagonal, U on and above.  H col-major n x n. */
static void lu_factor(int n, double* H, int* perm) {
#define Hc(r, c) H[(size_t)(c) * n + (r)]
    for (int i = 0; i < n; i++) perm[i] = i;
    for (int k = 0; k < n; k++) {
        int p = k; double mx = fabs(Hc(k, k));
        for (int r = k + 1; r < n; r++) { double v = fabs(Hc(r, k)); if (v > mx) { mx = v; p = r; } }
        if (p != k) {
            for (int c = 0; c < n; c++) { double t = Hc(k, c); Hc(k, c) = Hc(p, c); Hc(p, c) = t; }
            int t = perm[k]; perm[k] = perm[p]; perm[p] = t;
        }
        if (Hc(k, k) == 0.0) continue;
        double inv = 1.0 / Hc(k, k);
        for (int r = k + 1; r < n; r++) Hc(r, k) *= inv;
        for (int c = k + 1; c < n; c++) {
            double hkc = Hc(k, c);
            if (hkc != 0.0) for (int r = k + 1; r < n; r++) Hc(r, c) -= Hc(r, k) * hkc;
        }
    }
}
/* dx = -(Hu\(Hl\g(Hp)))  (:128) */
static void lu_apply_neg(int n, const double* LU, const int* perm, const double* g, double* dx) {
#define Lc(r, c) LU[(size_t)(c) * n + (r)]
    for (int i = 0; i < n; i++) dx[i] = g[perm[i]];
    for (int k = 0; k < n; k++) for (int r = k + 1; r < n; r++) dx[r] -= Lc(r, k) * dx[k];
    for (int k = n - 1; k >= 0; k--) { dx[k] /= Lc(k, k); for (int r = 0; r < k; r++) dx[r] -= Lc(r, k) * dx[k]; }
    for (int i = 0; i < n; i++) dx[i] = -dx[i];
}
/* zkk0(Hp) = Hl'\(Hu'\yk)   (TaskBDF1.m:76) */
static void lu_apply_transposed(int n, const double* LU, const int* perm, const double* y, double* z) {
    double* w = (double*)malloc(sizeof(double) * (size_t)(n + 1));
    for (int i = 0; i < n; i++) w[i] = y[i];
    for (int k = 0; k < n; k++) { w[k] /= Lc(k, k); for (int c = k + 1; c < n; c++) w[c] -= Lc(k, c) * w[k]; }     /* U' w = y */
    for (int k = n - 1; k >= 0; k--) for (int c = 0; c < k; c++) w[c] -= Lc(k, c) * w[k];                          /* L' u = w */
    for (int i = 0; i < n; i++) z[perm[i]] = w[i];
    free(w);
#undef Lc
#undef Hc
}

/* taskObjective (driverRedMaxAdjointBDF1.m:39-62) for TaskBDF1PointPos: scene.reset(), forward simLoop (:65-102) with the
 * line-search-free newton (:105-146; note x = x+dx BEFORE the |g|<tol test, so the stored factors/M/D/J belong to the last
 * EVALUATED iterate), TaskBDF1PointPos.applyStep/calcStep (TaskBDF1PointPos.m:58-107) and the backward sweep
 * TaskBDF1.calcFinal (TaskBDF1.m:45-81).  p: nr parameters (reduced order).  Returns P, fills dPdp (nr). */
double orc_adjoint_bdf1(orc_scene* s, double h, int nsteps, const orc_task_pointpos* task, const double* p, double* dPdp, orc_stats* st) {
    const int nr = s->nr, nm = s->nm;
    const size_t n2 = (size_t)nr * nr;
    double* LUh = (double*)calloc(n2 * (size_t)nsteps + 1, sizeof(double));
    double* Mh = (double*)calloc(n2 * (size_t)nsteps + 1, sizeof(double));
    double* Dh = (double*)calloc(n2 * (size_t)nsteps + 1, sizeof(double));
    int* Ph = (int*)calloc((size_t)nr * nsteps + 1, sizeof(int));
    double* dPdq = (double*)calloc((size_t)nr * nsteps + 1, sizeof(double));
    double *q0 = calloc((size_t)nr + 1, 8), *qd0 = calloc((size_t)nr + 1, 8), *x = calloc((size_t)nr + 1, 8), *qB = calloc((size_t)nr + 1, 8);
    double *g = calloc((size_t)nr + 1, 8), *dx = calloc((size_t)nr + 1, 8), *qd1 = calloc((size_t)nr + 1, 8);
    double *H = calloc(n2 + 1, 8), *M = calloc(n2 + 1, 8), *f = calloc((size_t)nr + 1, 8), *K = calloc(n2 + 1, 8), *D = calloc(n2 + 1, 8);
    double *dMdq = calloc(n2 * nr + 1, 8), *Jk = calloc((size_t)nm * nr + 1, 8);
    const double tol = 1e-9, dxMax = 1e3;           /* :106-107 */
    const int iterMax = 5 * nr;                      /* :108 */
    if (st) memset(st, 0, sizeof(*st));
    orc_reset(s);                                    /* scene.reset() :40 */
    double P = 0.0, t = 0.0;                         /* task.init(): P = 0 */
    for (int k = 1; k <= nsteps; k++) {
        /* task.applyStep(): joint.tau = pscale*p(idxR)   TaskBDF1PointPos.m:58-64 */
        for (int i = 0; i < s->n; i++) if (s->nd[i].ndof) s->nd[i].tau = task->pscale * p[s->nd[i].idxR];
        orc_get_state(s, q0, qd0);
        for (int i = 0; i < nr; i++) { x[i] = q0[i] + h * qd0[i]; qB[i] = q0[i] + h * qd0[i]; }
        double* LUk = LUh + n2 * (size_t)(k - 1); int* Pk = Ph + (size_t)nr * (k - 1);
        int iter = 1;
        while (1) {
            /* [g,H,M,f,K,D,J] = evalBDF1(x) :160-176 */
            for (int i = 0; i < nr; i++) qd1[i] = (x[i] - q0[i]) / h;
            set_q(s, x, qd1); scene_update(s);
            orc_compute_values(s, M, f, dMdq, K, D);
            memcpy(Jk, s->J, sizeof(double) * (size_t)nm * nr);
            for (int a = 0; a < nr; a++) { double tt = 0; for (int b = 0; b < nr; b++) tt += M[(size_t)b * nr + a] * (x[b] - qB[b]); g[a] = tt - h * h * f[a]; }
            for (size_t i = 0; i < n2; i++) H[i] = M[i] - h * D[i] - h * h * K[i];
            for (int i = 0; i < nr; i++) {
                const double* Di = dMdq + (size_t)i * n2;
                for (int a = 0; a < nr; a++) { double tt = 0; for (int b = 0; b < nr; b++) tt += Di[(size_t)b * nr + a] * (x[b] - qB[b]); H[(size_t)i * nr + a] += tt; }
            }
            if (st) { st->newton_iters++; st->hessian_evals++; }
            memcpy(LUk, H, sizeof(double) * n2);
            lu_factor(nr, LUk, Pk);                   /* :127 */
            lu_apply_neg(nr, LUk, Pk, g, dx);         /* :128 */
            if (vnorm(nr, dx) > dxMax) { if (st) st->diverged++; break; }
            for (int i = 0; i < nr; i++) x[i] += dx[i];   /* :134 (before the convergence test) */
            if (vnorm(nr, g) < tol) break;
            if (iter >= iterMax) { if (st) st->not_converged++; break; }
            iter++;
        }
        for (int i = 0; i < nr; i++) qd1[i] = (x[i] - q0[i]) / h;
        set_q(s, x, qd1); scene_update(s);
        t += h;
        /* saveHistory(Hl,Hu,Hp,M,f,K,D,J) Scene.m:139-153 */
        memcpy(Mh + n2 * (size_t)(k - 1), M, sizeof(double) * n2);
        memcpy(Dh + n2 * (size_t)(k - 1), D, sizeof(double) * n2);
        /* task.calcStep()  TaskBDF1PointPos.m:67-107 */
        if (fabs(task->t - t) < 1e-6) {
            const onode* b = &s->nd[task->body];
            double xw[3], dxw[3];
            for (int a = 0; a < 3; a++) {
                xw[a] = b->E_wi[a][0] * task->xlocal[0] + b->E_wi[a][1] * task->xlocal[1] + b->E_wi[a][2] * task->xlocal[2] + b->E_wi[a][3];
                dxw[a] = xw[a] - task->xtarget[a];
            }
            P += task->wpos * 0.5 * (dxw[0] * dxw[0] + dxw[1] * dxw[1] + dxw[2] * dxw[2]);
            /* dxdqm(:,idxM) = R*Gamma(xlocal), Gamma = [brac(xlocal)', I]; dPdq = J'*dxdqm'*dx*wp */
            double G[3][6], xb[3][3], RG[3][6];
            se3_brac3(xb, task->xlocal);
            for (int a = 0; a < 3; a++) for (int c = 0; c < 3; c++) { G[a][c] = xb[c][a]; G[a][3 + c] = (a == c); }
            for (int a = 0; a < 3; a++) for (int c = 0; c < 6; c++) { double tt = 0; for (int e = 0; e < 3; e++) tt += b->E_wi[a][e] * G[e][c]; RG[a][c] = tt; }
            double wm[6];
            for (int c = 0; c < 6; c++) wm[c] = (RG[0][c] * dxw[0] + RG[1][c] * dxw[1] + RG[2][c] * dxw[2]) * task->wpos;
            for (int a = 0; a < nr; a++) {
                double tt = 0;
                for (int c = 0; c < 6; c++) tt += Jk[(size_t)a * nm + b->idxM + c] * wm[c];
                dPdq[(size_t)nr * (k - 1) + a] = tt;
            }
        }
    }
    /* TaskBDF1.calcFinal  TaskBDF1.m:45-81 */
    double wreg2 = 0; for (int i = 0; i < nr; i++) wreg2 += p[i] * p[i];
    P += task->wreg * 0.5 * wreg2;
    double* z = (double*)calloc((size_t)nr * (nsteps + 2) + 1, sizeof(double));
    double* yk = (double*)calloc((size_t)nr + 1, sizeof(double));
    for (int k = nsteps; k >= 1; k--) {
        for (int a = 0; a < nr; a++) yk[a] = dPdq[(size_t)nr * (k - 1) + a];
        if (k < nsteps) {      /* block = -2*M + h*D of step k+1; yk -= block'*z(k+1) */
            const double* M1 = Mh + n2 * (size_t)k; const double* D1 = Dh + n2 * (size_t)k; const double* z1 = z + (size_t)nr * k;
            for (int a = 0; a < nr; a++) { double tt = 0; for (int j = 0; j < nr; j++) tt += (-2.0 * M1[(size_t)a * nr + j] + h * D1[(size_t)a * nr + j]) * z1[j]; yk[a] -= tt; }
        }
        if (k < nsteps - 1) {  /* block = M of step k+2 */
            const double* M2 = Mh + n2 * (size_t)(k + 1); const double* z2 = z + (size_t)nr * (k + 1);
            for (int a = 0; a < nr; a++) { double tt = 0; for (int j = 0; j < nr; j++) tt += M2[(size_t)a * nr + j] * z2[j]; yk[a] -= tt; }
        }
        lu_apply_transposed(nr, LUh + n2 * (size_t)(k - 1), Ph + (size_t)nr * (k - 1), yk, z + (size_t)nr * (k - 1));
    }
    /* dPdp = wreg*p' - z'*dgdp, dgdp(kk,:) = -h^2*pscale*I  (TaskBDF1PointPos.m:104-105) */
    for (int a = 0; a < nr; a++) {
        double zs = 0; for (int k = 0; k < nsteps; k++) zs += z[(size_t)nr * k + a];
        dPdp[a] = task->wreg * p[a] + h * h * task->pscale * zs;
    }
    for (int i = 0; i < s->n; i++) s->nd[i].tau = 0.0;
    free(LUh); free(Mh); free(Dh); free(Ph); free(dPdq); free(q0); free(qd0); free(x); free(qB); free(g); free(dx); free(qd1);
    free(H); free(M); free(f); free(K); free(D); free(dMdq); free(Jk); free(z); free(yk);
    return P;
}

/* taskObjective (driverRedMaxAdjointBDF2.m:38-62) for TaskBDF2PointPos (scene 101, scenesRedMax.m:437-471): scene.reset(), the
 * forward simLoop (:65-136: SDIRK2a + SDIRK2b on the first step, BDF2 afterwards) with the line-search-free newton (:139-181, x = x+dx
 * BEFORE the |g| < tol test, iterMax = 5 nr), saveHistory of the factors of H and of M, D, J of the LAST evaluated iterate of the
 * step's final solve (step 1: SDIRK2b), TaskBDF2PointPos.applyStep / calcStep (TaskBDF2PointPos.m:58-108) and the backward sweep
 * TaskBDF2.calcFinal (TaskBDF2.m:45-107) with its four off-diagonal blocks.  Every solve is the common residual
 *     qdot = (x - qA)/eta,  dqtmp = x - qB,  g = M dqtmp - eta^2 f,  H = M - eta D - eta^2 K + dMdq dqtmp
 * (evalSDIRK2a :184-216, evalSDIRK2b :219-252, evalBDF2 :255-287).  As in the reference, dg/dp = -(4/9) h^2 pscale I is used for
 * EVERY step (TaskBDF2PointPos.m:97-106), the SDIRK start step included, and dg/dqa is dropped (TaskBDF2.m:52-55).  p: nr
 * parameters (reduced order).  Returns P, fills dPdp (nr). */
double orc_adjoint_bdf2(orc_scene* s, double h, int nsteps, const orc_task_pointpos* task, const double* p, double* dPdp, orc_stats* st) {
    const int nr = s->nr, nm = s->nm;
    const size_t n2 = (size_t)nr * nr;
    double* LUh = (double*)calloc(n2 * (size_t)nsteps + 1, sizeof(double));
    double* Mh = (double*)calloc(n2 * (size_t)nsteps + 1, sizeof(double));
    double* Dh = (double*)calloc(n2 * (size_t)nsteps + 1, sizeof(double));
    int* Ph = (int*)calloc((size_t)nr * nsteps + 1, sizeof(int));
    double* dPdq = (double*)calloc((size_t)nr * nsteps + 1, sizeof(double));
    const size_t sz = (size_t)nr + 1;
    double *q0 = calloc(sz, 8), *qd0 = calloc(sz, 8), *q1 = calloc(sz, 8), *qd1 = calloc(sz, 8), *qa = calloc(sz, 8), *qda = calloc(sz, 8);
    double *x = calloc(sz, 8), *qA = calloc(sz, 8), *qB = calloc(sz, 8), *xd = calloc(sz, 8), *g = calloc(sz, 8), *dx = calloc(sz, 8);
    double *H = calloc(n2 + 1, 8), *M = calloc(n2 + 1, 8), *f = calloc(sz, 8), *K = calloc(n2 + 1, 8), *D = calloc(n2 + 1, 8);
    double *dMdq = calloc(n2 * nr + 1, 8), *Jk = calloc((size_t)nm * nr + 1, 8), *LUs = calloc(n2 + 1, 8);
    int* Ps = (int*)calloc(sz, sizeof(int));
    const double tol = 1e-9, dxMax = 1e3;           /* :140-141 */
    const int iterMax = 5 * nr;                      /* :142 */
    const double al = (2.0 - sqrt(2.0)) / 2.0;
    if (st) memset(st, 0, sizeof(*st));
    orc_reset(s);
    double P = 0.0, t = 0.0;
    for (int k = 1; k <= nsteps; k++) {
        for (int i = 0; i < s->n; i++) if (s->nd[i].ndof) s->nd[i].tau = task->pscale * p[s->nd[i].idxR];      /* applyStep */
        double* LUk = LUh + n2 * (size_t)(k - 1); int* Pk = Ph + (size_t)nr * (k - 1);
        const int nsolve = k == 1 ? 2 : 1;
        for (int sv = 0; sv < nsolve; sv++) {
            double eta;
            if (k == 1 && sv == 0) {            /* SDIRK2a :78-84 */
                orc_get_state(s, q0, qd0);
                eta = al * h;
                for (int i = 0; i < nr; i++) { x[i] = q0[i] + al * h * qd0[i]; qA[i] = q0[i]; qB[i] = q0[i] + (al * h) * qd0[i]; }
            } else if (k == 1) {                /* SDIRK2b :89-92 */
                eta = al * h;
                for (int i = 0; i < nr; i++) {
                    x[i] = qa[i] + (1 - al) * h * qda[i];
                    qA[i] = q0[i] + (1 - al) * h * qda[i];
                    qB[i] = q0[i] + (2 * al - 1) * h * qd0[i] + 2 * (1 - al) * h * qda[i];
                }
            } else {                            /* BDF2 :103-113 */
                for (int i = 0; i < s->n; i++) if (s->nd[i].ndof) { q0[s->nd[i].idxR] = s->nd[i].q1; qd0[s->nd[i].idxR] = s->nd[i].qdot1; }
                orc_get_state(s, q1, qd1);
                for (int i = 0; i < s->n; i++) { s->nd[i].q1 = s->nd[i].q; s->nd[i].qdot1 = s->nd[i].qdot; }
                eta = (2.0 / 3.0) * h;
                for (int i = 0; i < nr; i++) {
                    x[i] = q1[i] + h * qd1[i];
                    qA[i] = (4.0 / 3.0) * q1[i] - (1.0 / 3.0) * q0[i];
                    qB[i] = (4.0 / 3.0) * q1[i] - (1.0 / 3.0) * q0[i] + (8.0 / 9.0) * h * qd1[i] - (2.0 / 9.0) * h * qd0[i];
                }
            }
            const int keep = !(k == 1 && sv == 0);      /* the SDIRK2a solve leaves nothing in the history */
            double* LUx = keep ? LUk : LUs; int* Px = keep ? Pk : Ps;
            int iter = 1;
            while (1) {
                for (int i = 0; i < nr; i++) xd[i] = (x[i] - qA[i]) / eta;
                set_q(s, x, xd); scene_update(s);
                orc_compute_values(s, M, f, dMdq, K, D);
                memcpy(Jk, s->J, sizeof(double) * (size_t)nm * nr);
                for (int a = 0; a < nr; a++) { double tt = 0; for (int b = 0; b < nr; b++) tt += M[(size_t)b * nr + a] * (x[b] - qB[b]); g[a] = tt - eta * eta * f[a]; }
                for (size_t i = 0; i < n2; i++) H[i] = M[i] - eta * D[i] - eta * eta * K[i];
                for (int i = 0; i < nr; i++) {
                    const double* Di = dMdq + (size_t)i * n2;
                    for (int a = 0; a < nr; a++) { double tt = 0; for (int b = 0; b < nr; b++) tt += Di[(size_t)b * nr + a] * (x[b] - qB[b]); H[(size_t)i * nr + a] += tt; }
                }
                if (st) { st->newton_iters++; st->hessian_evals++; }
                memcpy(LUx, H, sizeof(double) * n2);
                lu_factor(nr, LUx, Px);
                lu_apply_neg(nr, LUx, Px, g, dx);
                if (vnorm(nr, dx) > dxMax) { if (st) st->diverged++; break; }
                for (int i = 0; i < nr; i++) x[i] += dx[i];
                if (vnorm(nr, g) < tol) break;
                if (iter >= iterMax) { if (st) st->not_converged++; break; }
                iter++;
            }
            if (k == 1 && sv == 0) {
                for (int i = 0; i < nr; i++) { qa[i] = x[i]; qda[i] = (x[i] - q0[i]) / (al * h); }
            } else if (k == 1) {
                for (int i = 0; i < nr; i++) xd[i] = (x[i] - q0[i] - (1 - al) * h * qda[i]) / (al * h);
                set_q(s, x, xd);
                for (int i = 0; i < s->n; i++) if (s->nd[i].ndof) { s->nd[i].q1 = q0[s->nd[i].idxR]; s->nd[i].qdot1 = qd0[s->nd[i].idxR]; }
            } else {
                for (int i = 0; i < nr; i++) xd[i] = (3.0 / (2.0 * h)) * (x[i] - (4.0 / 3.0) * q1[i] + (1.0 / 3.0) * q0[i]);
                set_q(s, x, xd);
            }
        }
        scene_update(s);
        t += h;
        memcpy(Mh + n2 * (size_t)(k - 1), M, sizeof(double) * n2);
        memcpy(Dh + n2 * (size_t)(k - 1), D, sizeof(double) * n2);
        if (fabs(task->t - t) < 1e-6) {         /* TaskBDF2PointPos.calcStep :67-95 */
            const onode* b = &s->nd[task->body];
            double xw[3], dxw[3];
            for (int a = 0; a < 3; a++) {
                xw[a] = b->E_wi[a][0] * task->xlocal[0] + b->E_wi[a][1] * task->xlocal[1] + b->E_wi[a][2] * task->xlocal[2] + b->E_wi[a][3];
                dxw[a] = xw[a] - task->xtarget[a];
            }
            P += task->wpos * 0.5 * (dxw[0] * dxw[0] + dxw[1] * dxw[1] + dxw[2] * dxw[2]);
            double G[3][6], xb[3][3], RG[3][6];
            se3_brac3(xb, task->xlocal);
            for (int a = 0; a < 3; a++) for (int c = 0; c < 3; c++) { G[a][c] = xb[c][a]; G[a][3 + c] = (a == c); }
            for (int a = 0; a < 3; a++) for (int c = 0; c < 6; c++) { double tt = 0; for (int e = 0; e < 3; e++) tt += b->E_wi[a][e] * G[e][c]; RG[a][c] = tt; }
            double wm[6];
            for (int c = 0; c < 6; c++) wm[c] = (RG[0][c] * dxw[0] + RG[1][c] * dxw[1] + RG[2][c] * dxw[2]) * task->wpos;
            for (int a = 0; a < nr; a++) {
                double tt = 0;
                for (int c = 0; c < 6; c++) tt += Jk[(size_t)a * nm + b->idxM + c] * wm[c];
                dPdq[(size_t)nr * (k - 1) + a] = tt;
            }
        }
    }
    /* TaskBDF2.calcFinal  TaskBDF2.m:45-107 */
    double wreg2 = 0; for (int i = 0; i < nr; i++) wreg2 += p[i] * p[i];
    P += task->wreg * 0.5 * wreg2;
    double* z = (double*)calloc((size_t)nr * (nsteps + 5) + 1, sizeof(double));
    double* yk = (double*)calloc(sz, sizeof(double));
    for (int k = nsteps; k >= 1; k--) {
        for (int a = 0; a < nr; a++) yk[a] = dPdq[(size_t)nr * (k - 1) + a];
        for (int j = 1; j <= 4; j++) {
            if (!(k < nsteps - (j - 1))) continue;
            double cm, cd;                       /* block = cm M + cd h D of step k + j   (:66-96) */
            if (j == 1) { cm = (k == 1) ? -((8.0 / (9.0 * al)) + (4.0 / 3.0)) : -(8.0 / 3.0); cd = 8.0 / 9.0; }
            else if (j == 2) { cm = (k == 1) ? ((2.0 / (9.0 * al)) + (19.0 / 9.0)) : (22.0 / 9.0); cd = -2.0 / 9.0; }
            else if (j == 3) { cm = -8.0 / 9.0; cd = 0.0; }
            else { cm = 1.0 / 9.0; cd = 0.0; }
            const double* Mj = Mh + n2 * (size_t)(k + j - 1); const double* Dj = Dh + n2 * (size_t)(k + j - 1); const double* zj = z + (size_t)nr * (k + j - 1);
            for (int a = 0; a < nr; a++) { double tt = 0; for (int c = 0; c < nr; c++) tt += (cm * Mj[(size_t)a * nr + c] + cd * h * Dj[(size_t)a * nr + c]) * zj[c]; yk[a] -= tt; }
        }
        lu_apply_transposed(nr, LUh + n2 * (size_t)(k - 1), Ph + (size_t)nr * (k - 1), yk, z + (size_t)nr * (k - 1));
    }
    for (int a = 0; a < nr; a++) {               /* dPdp = wreg*p' - z'*dgdp, dgdp(kk,:) = -(4/9) h^2 pscale I */
        double zs = 0; for (int k = 0; k < nsteps; k++) zs += z[(size_t)nr * k + a];
        dPdp[a] = task->wreg * p[a] + (4.0 / 9.0) * h * h * task->pscale * zs;
    }
    for (int i = 0; i < s->n; i++) s->nd[i].tau = 0.0;
    free(LUh); free(Mh); free(Dh); free(Ph); free(dPdq); free(q0); free(qd0); free(q1); free(qd1); free(qa); free(qda);
    free(x); free(qA); free(qB); free(xd); free(g); free(dx); free(H); free(M); free(f); free(K); free(D); free(dMdq); free(Jk); free(LUs); free(Ps);
    free(z); free(yk);
    return P;
}

/* -------------------------------------------------- batch CPU baseline */

long orc_batch_step_bdf1_ex2(const orc_desc* d, int B, double* q, double* qdot, double h, int nsteps, int nthreads, int* iters, int* halvings, int* bad,
                             int* diverged, double* worst_exit_g);
long orc_batch_step_bdf1_ex(const orc_desc* d, int B, double* q, double* qdot, double h, int nsteps, int nthreads, int* iters, int* halvings, int* bad) {
    return orc_batch_step_bdf1_ex2(d, B, q, qdot, h, nsteps, nthreads, iters, halvings, bad, NULL, NULL);
}
long orc_batch_step_bdf1(const orc_desc* d, int B, double* q, double* qdot, double h, int nsteps, int nthreads) {
    return orc_batch_step_bdf1_ex(d, B, q, qdot, h, nsteps, nthreads, NULL, NULL, NULL);
}
/* the same with per-rollout counters: Newton iterations, line-search halvings, diverged + not-converged steps ([B] or NULL) */
/* diverged: "Newton diverged" steps alone ([B] or NULL); worst_exit_g: the largest |g| a not-converged step of the rollout ended with */
long orc_batch_step_bdf1_ex2(const orc_desc* d, int B, double* q, double* qdot, double h, int nsteps, int nthreads, int* iters, int* halvings, int* bad,
                             int* diverged, double* worst_exit_g) {
    long total = 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel reduction(+ : total)
    {
        orc_scene* s = orc_create(d);
        int nr = s->nr;
        double* qrest = (double*)calloc((size_t)nr + 1, sizeof(double));
        orc_get_state(s, qrest, NULL);       /* model constant: qRest = descriptor q */
#pragma omp for schedule(dynamic, 1)
        for (int b = 0; b < B; b++) {
            orc_stats st;
            orc_set_state(s, q + (size_t)b * nr, qdot + (size_t)b * nr);
            orc_set_qrest(s, qrest);
            orc_step_bdf1(s, h, nsteps, &st, NULL, NULL);
            orc_get_state(s, q + (size_t)b * nr, qdot + (size_t)b * nr);
            total += st.newton_iters;
            if (iters) iters[b] = st.newton_iters;
            if (halvings) halvings[b] = st.ls_halvings;
            if (bad) bad[b] = st.diverged + st.not_converged;
            if (diverged) diverged[b] = st.diverged;
            if (worst_exit_g) worst_exit_g[b] = st.worst_exit_g;
        }
        free(qrest);
        orc_destroy(s);
    }
    return total;
}
