/*
 * redmax_oracle.h -- CPU restatement of the sueda/redmax matlab-diff BDF1/BDF2 path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and
 * only as the checker / the reported CPU baseline.  The product (redmax_amd/,
 * include/redmax_hip.h) never links, imports or falls back to this code.
 *
 * It is a literal, single-trajectory, fp64 restatement of the reference's algorithm
 * (same data flow as the .m files, incl. the dense dJ/dq tensors and the O(n^3)
 * ancestor loops); every function cites the reference file:line it follows.
 * Parity pin: the reference's known-answer energies Hexpected(BDF1/BDF2) for scenes
 * 0,1,2,3,14 (matlab-diff/scenesRedMax.m:54-55,82-83,108-109,133-134,373-374) and
 * Hexpected(REDMAX_EULER) for matlab-simple scenes 0,1,2 (matlab/testRedMaxScenes.m:39,67,93)
 * -- see tests/test_oracle_kat.py.  MATLAB/Octave are not available, so the
 * reference itself cannot be executed here.
 */
#ifndef REDMAX_ORACLE_H
#define REDMAX_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_JOINT_FIXED = 0, ORC_JOINT_REVOLUTE = 1, ORC_JOINT_PRISMATIC = 2 };

/* Flat scene listing: joints/bodies in the order the reference scene file lists them
 * (parent before child, scenesRedMax.m).  All 4x4 transforms are COLUMN-MAJOR (MATLAB
 * memory layout), 16 doubles each. */
typedef struct orc_desc {
    int njoints;
    const int* parent;       /* [n]  index of parent joint, -1 for the root              */
    const int* type;         /* [n]  ORC_JOINT_*                                         */
    const double* axis;      /* [n][3] joint axis (normalised inside, JointRevolute.m:14) */
    const double* E0_pj;     /* [n][16] setJointTransform(E)  (Joint.m:95-99)            */
    const double* E0_ji;     /* [n][16] setBodyTransform(E)   (Body.m:46-51)             */
    const double* I_i;       /* [n][6]  body inertia diag (se3.inertiaCuboid)            */
    const double* q;         /* [n] initial joint position (ignored for FIXED)           */
    const double* qdot;      /* [n] initial joint velocity                               */
    const double* tau;       /* [n] */
    const double* stiffness; /* [n] */
    const double* damping;   /* [n] */
    const double* qLimL;     /* [n] */
    const double* qLimU;     /* [n] */
    const double* qLimK;     /* [n] */
    const double* qLimD;     /* [n] */
    double grav[3];
    int normalize_axis;      /* 1 = matlab-diff (JointRevolute.m:14), 0 = matlab-simple  */
} orc_desc;

typedef struct orc_scene orc_scene;

orc_scene* orc_create(const orc_desc* d);               /* Scene.init()  Scene.m:59-119 */
void orc_destroy(orc_scene* s);
int  orc_nr(const orc_scene* s);
int  orc_nm(const orc_scene* s);
void orc_idxR(const orc_scene* s, int* idx);            /* [n] 0-based start of each joint's reduced index, -1 if ndof=0 */
/* JointSpherical / JointFree3D (JointSpherical.m, JointFree3D.m) held as three revolute nodes per joint: first[g] = the first
 * node of group g.  Charts are the reference's CHART_* numbers 1..12 (XYX XZX YZY YXY ZXZ ZYZ XYZ XZY YZX YXZ ZXY ZYX),
 * 7 = XYZ at construction; orc_step_bdf1/bdf2 run reparam_ (:63-102) after every step. */
int  orc_set_spherical(orc_scene* s, int ngroups, const int* first);
void orc_get_charts(const orc_scene* s, int* charts);
int  orc_set_charts(orc_scene* s, const int* charts);
double orc_euler(int chart, const double* q, double* R9, double* T9);   /* getEuler: R, T row-major; returns det T */
void orc_euler_inv(int chart, const double* R9, double* q);             /* getEulerInv */
int  orc_set_idxR(orc_scene* s, const int* idx);        /* explicit reduced numbering (lowered multi-DOF joints) */
void orc_reset(orc_scene* s);                           /* Scene.reset() Scene.m:122-131 */
void orc_get_state(const orc_scene* s, double* q, double* qdot);   /* Joint.getQ  */
void orc_set_state(orc_scene* s, const double* q, const double* qdot); /* Joint.setQ + update() */
void orc_set_qrest(orc_scene* s, const double* qrest);  /* override qRest (reduced order) */
void orc_energy(const orc_scene* s, double* T, double* V);          /* Joint.computeEnergies */

/* ForceGroundCuboid (matlab-diff/+redmax/ForceGroundCuboid.m): penalty ground contact with friction on the 8 corners of the
 * flagged cuboids.  flags[n], sides[n][3], ground frame E (column-major 4x4, Z up), setStiffness(kn,kt), setFriction(mu),
 * setDamping(kd).  Pinned by Hexpected of scene 11 (scenesRedMax.m:292-293) with JointFree2D emulated by a
 * prismatic-x / prismatic-y / revolute-z chain of massless links (same reduced coordinates, JointFree2D.m:20-33). */
void orc_set_ground_contact(orc_scene* s, const int* flags, const double* sides, const double* E16, double kn, double kt, double mu, double kd);
/* the same with one frame / one set of constants PER BODY (every ForceGroundCuboid object holds its own E, kn, kt, mu, kd):
 * E16 [n][16] if E_per_body else [16]; kn..kd [n] if k_per_body else [1] */
void orc_set_ground_contact_body(orc_scene* s, const int* flags, const double* sides, const double* E16, int E_per_body,
                                 const double* kn, const double* kt, const double* mu, const double* kd, int k_per_body);
/* a further ForceGroundCuboid on an already flagged body (the reference's force list has no one-per-body rule); -1: body not flagged */
int orc_add_ground_contact(orc_scene* s, int body, const double* E16, double kn, double kt, double mu, double kd);

/* Joint.computeJacobian at the current state; any pointer may be NULL.
 * J,Jdot: nm x nr column-major; dJdq,dJdotdq: nm x nr x nr (MATLAB layout). */
void orc_jacobian(orc_scene* s, double* J, double* Jdot, double* dJdq, double* dJdotdq);

/* computeValues (driverRedMaxBDF1.m:190-243) at the current state.
 * M,K,D: nr x nr col-major; f: nr; dMdq: nr x nr x nr. K, D, dMdq may be NULL (2-output form). */
void orc_compute_values(orc_scene* s, double* M, double* f, double* dMdq, double* K, double* D);

/* Generic implicit residual shared by evalBDF1 / evalSDIRK2a / evalSDIRK2b / evalBDF2:
 *   qdot = (q - qA)/eta ; dqtmp = q - qB ; g = M*dqtmp - eta^2 f ; H = M - eta*D - eta^2*K + dMdq*dqtmp
 * BDF1 (driverRedMaxBDF1.m:160-187): eta=h, qA=q0, qB=q0+h*qdot0.  H may be NULL (g-only path). */
void orc_eval_residual(orc_scene* s, const double* q, const double* qA, const double* qB,
                       double eta, double* g, double* H);

typedef struct orc_stats {
    int newton_iters;     /* total Newton iterations                                  */
    int ls_halvings;      /* total line-search halvings (iterLs-1 summed)             */
    int residual_evals;   /* number of g-only evaluations                             */
    int hessian_evals;    /* number of (g,H) evaluations                              */
    int diverged;         /* steps that hit "Newton diverged"                         */
    int not_converged;    /* steps that hit iterMax                                   */
    int chart_switches;   /* JointSpherical.reparam_ chart changes                    */
    double worst_exit_g;  /* largest |g| a "did not converge" step ended with (0 if none): tells a stall at the resolution of
                             doubles (|g| ~ tol) from a step that is really lost                                           */
} orc_stats;

/* Newton constants used by every orc_step_* call (process-global).  Defaults = the reference's hard-coded
 * values tol=1e-9, dxMax=1e3, iterMax=10*nr, iterLsMax=20 (driverRedMaxBDF1.m:95-98). */
/* diagnostic (single-threaded): newton() logs {|g| at the start of an iteration, |g| after its line search, trials} per iteration */
void orc_set_trace(double* buf, int cap_iters);
int orc_trace_count(void);
void orc_set_newton(double tol, double dxMax, int iterMaxPerDof, int iterLsMax);
/* rmx_opts.ls_fail_limit restated (NOT reference behaviour; 0 = off = the reference) */
void orc_set_ls_fail_limit(int n);

/* driverRedMaxBDF1.m simLoop:57-91 + newton:94-157.  Advances nsteps steps.
 * If Hist_T/Hist_V non-NULL they receive T,V after each step (Scene.saveHistory). */
void orc_step_bdf1(orc_scene* s, double h, int nsteps, orc_stats* st, double* Hist_T, double* Hist_V);

/* driverRedMaxBDF2.m simLoop:57-125 (SDIRK2 start, then BDF2). step0 = index of the first step
 * to take (0 => take the SDIRK2 start step first). */
void orc_step_bdf2(orc_scene* s, double h, int step0, int nsteps, orc_stats* st, double* Hist_T, double* Hist_V);

/* matlab-simple/testRedMax.m euler:67-109 (linearly-implicit Euler, config 1). */
void orc_step_euler_simple(orc_scene* s, double h, int nsteps, double* Hist_T, double* Hist_V);

/* TaskBDF1PointPos (matlab-diff/+redmax/TaskBDF1PointPos.m): move a point of a body to a target at time t; the
 * parameters are constant joint torques tau = pscale*p. */
typedef struct orc_task_pointpos {
    int body;                 /* listing index of the body                         setBody   */
    double xlocal[3];         /* point in body coordinates                         setPoint  */
    double xtarget[3];        /* world target                                      setTarget */
    double t;                 /* measurement time                                  setTime   */
    double pscale;            /* torque scale                                      setScale  */
    double wreg, wpos;        /* regulariser / position weights                    setWeights */
} orc_task_pointpos;

/* taskObjective of driverRedMaxAdjointBDF1.m:39-62: reset, forward sim storing LU factors/M/D/J per step, backward sweep
 * TaskBDF1.calcFinal.  Returns P, fills dPdp[nr].  Pinned only by the finite-difference identity (testGrad :46-61): the
 * reference holds no golden numbers for this path ("parity unpinned" beyond FD). */
double orc_adjoint_bdf1(orc_scene* s, double h, int nsteps, const orc_task_pointpos* task, const double* p, double* dPdp, orc_stats* st);
/* the same for driverRedMaxAdjointBDF2.m / TaskBDF2.m / TaskBDF2PointPos.m (scene 101) */
double orc_adjoint_bdf2(orc_scene* s, double h, int nsteps, const orc_task_pointpos* task, const double* p, double* dPdp, orc_stats* st);

/* Batch helper for the timed CPU baseline: B independent trajectories of the same scene,
 * OpenMP over trajectories (nthreads), q/qdot [B][nr] in/out. Returns total Newton iterations. */
long orc_batch_step_bdf1(const orc_desc* d, int B, double* q, double* qdot, double h, int nsteps, int nthreads);
/* the same, also returning per-rollout counters ([B] each, any may be NULL): Newton iterations, line-search halvings,
 * number of steps that ended in "Newton diverged" / "did not converge" */
long orc_batch_step_bdf1_ex(const orc_desc* d, int B, double* q, double* qdot, double h, int nsteps, int nthreads,
                            int* iters, int* halvings, int* bad);
/* ... and, separately, the "Newton diverged" steps alone and the largest |g| a "did not converge" step of the rollout ended with
 * (tools/max_valid_amplitude.py: which initial-state ranges the reference algorithm itself survives) */
long orc_batch_step_bdf1_ex2(const orc_desc* d, int B, double* q, double* qdot, double h, int nsteps, int nthreads,
                             int* iters, int* halvings, int* bad, int* diverged, double* worst_exit_g);

/* ---- redmax_tensorfree.c: the tensor-free CPU baseline ("Baseline B"): the algorithm the HIP kernels
 * execute (world-frame recursive Newton-Euler + analytic derivatives), scalar C, same Newton, OpenMP over rollouts.
 * Trees of fixed / revolute / prismatic joints, no contact.  Checked against the literal restatement above. */
int  otf_nr(const orc_desc* d);
void otf_eval(const orc_desc* d, const double* q, const double* qA, const double* qB, double eta, double* g, double* H);
/* the same at the compensated iterate q + qlo (|qlo| <= ulp(q)/2): qlo enters v = q - qB and qdot = (q - qA)/eta only */
void otf_eval_lo(const orc_desc* d, const double* q, const double* qlo, const double* qA, const double* qB, double eta, double* g, double* H);
long otf_batch_step_bdf1(const orc_desc* d, int B, double* q, double* qdot, double h, int nsteps, int nthreads, double tol,
                         double dxMax, int iterMaxPerDof, int iterLsMax, int compensated, int* iters, int* halvings, int* status);

#ifdef __cplusplus
}
#endif
#endif
