/*
 * redmax_hip.h -- C ABI of the MI355X-native batched RedMax BDF1/BDF2 forward-dynamics step.
 *
 * The reference (sueda/redmax) has no plugin / FFI / MEX boundary for this path: everything runs
 * inside one MATLAB interpreter.  The seam this library replaces is the body of
 *     simLoop(scene)                     matlab-diff/driverRedMaxBDF1.m:57-91
 *       newton(@(q1)evalBDF1(q1,scene))  driverRedMaxBDF1.m:94-157, 160-187
 *         computeValues(scene)           driverRedMaxBDF1.m:190-243
 *           Joint.update / computeJacobian / computeForce   +redmax/Joint.m:382-613
 *           Body.update / computeMassGrav                   +redmax/Body.m:70-135
 *           se3.Ad / ad / inv / aaToMat                     se3.m:11-176
 * re-expressed as a batch over independent trajectories of ONE scene.  Each entry point cites the
 * reference code it stands in for.  A MATLAB MEX gateway or any other host binds
 * these symbols; the in-repo host is the ctypes mirror in redmax_amd/.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes only.  No exceptions cross the boundary.
 *   - every function returns 0 on success, a negative RMX_E_* code otherwise; rmx_last_error()
 *     gives the text.  Numerical failure (Newton diverged / not converged) is NOT an error: like
 *     the reference (driverRedMaxBDF1.m:118-121,150-153: print and continue) it is reported per
 *     trajectory in the stats arrays and stepping continues.
 *   - all 4x4 transforms are COLUMN-MAJOR (what MATLAB's mxGetPr returns for a 4x4 double).
 *   - reduced vectors use the reference's leaf-to-root DOF numbering (Scene.m:65-71): the LAST
 *     listed joint owns index 0.  Batched arrays are [batch][nr], trajectory-major, fp64.
 *   - matrices returned by rmx_eval are nr x nr COLUMN-MAJOR per trajectory ([batch][nr*nr]).
 *   - host pointers are copied in/out and never retained; the library owns device memory.
 *   - there is NO CPU fallback: without a usable HIP device every create call fails loudly.
 */
#ifndef REDMAX_HIP_H
#define REDMAX_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define RMX_VERSION 111

enum {
    RMX_OK = 0,
    RMX_E_INVALID = -1,     /* bad argument / unsupported scene                        */
    RMX_E_NODEVICE = -2,    /* no HIP device (there is no CPU fallback)                */
    RMX_E_HIP = -3,         /* HIP runtime error, see rmx_last_error()                 */
    RMX_E_NOMEM = -4
};

enum {
    RMX_JOINT_FIXED = 0,          /* JointFixed.m                                                              */
    RMX_JOINT_REVOLUTE = 1,       /* JointRevolute.m    (axis)                                                 */
    RMX_JOINT_PRISMATIC = 2,      /* JointPrismatic.m   (axis)                                                 */
    RMX_JOINT_PLANAR = 3,         /* JointPlanar.m        2 DOF, p = B q            (plane, default x-y)       */
    RMX_JOINT_TRANSLATIONAL = 4,  /* JointTranslational.m 3 DOF, p = q                                         */
    RMX_JOINT_UNIVERSAL = 5,      /* JointUniversal.m     2 DOF, R = X(q1) Y(q2)                               */
    RMX_JOINT_FREE2D = 6,         /* JointFree2D.m        3 DOF, Q = [Rz(q3) [q1;q2;0]]                        */
    RMX_JOINT_SPHERICAL = 7,      /* JointSpherical.m     3 DOF, Euler angles in 12 switching charts           */
    RMX_JOINT_FREE3D = 8          /* JointFree3D.m        6 DOF, JointTranslational then JointSpherical        */
};

/* rmx_stats.status bits */
/* RMX_ST_STALLED accompanies RMX_ST_MAXITER when the Newton iteration reached a floating-point fixed point
 * (alpha*dx below one ulp of x): the reference repeats that identical iteration until iterMax and keeps x;
 * the library returns the same x without spinning. */
/* RMX_ST_PIVOTED is informational: at least one linear solve tripped the diagonal-pivot growth guard and was redone
 * with full partial pivoting (see rmx_device.h lu_solve_neg_diag). */
enum { RMX_ST_DIVERGED = 1, RMX_ST_MAXITER = 2, RMX_ST_NAN = 4, RMX_ST_STALLED = 8, RMX_ST_PIVOTED = 16,
       RMX_ST_CHART = 32 /* informational: a JointSpherical changed its Euler chart (the reference prints 'XYZ->YXZ') */,
       RMX_ST_LS_CUT = 128 /* with RMX_ST_MAXITER: rmx_opts.ls_fail_limit ended the Newton loop of a step (off by default) */,
       RMX_ST_COOP_FAULT = 512 /* with RMX_ST_NAN, chains with ForceGroundCuboid only: the wavefronts that share a creeping line search
                                  (DESIGN.md "Chains with ground contact") waited in vain for each other, or a rollout handed to them was
                                  never picked up; the rollout's state is NaN.  Never observed: a guard, not a code path of the method.
                                  (Bits 64 and 256 are internal to the kernels and never reach rmx_stats.status.) */ };

/* Scene listing, one entry per joint/body pair in the order the scene file lists them
 * (parent before child; scenesRedMax.m).  Replaces the handle-object graph that Scene.init()
 * links up (Scene.m:59-119).  Arrays are [njoints] unless noted. */
typedef struct rmx_model_desc {
    int njoints;
    const int* parent;       /* parent joint index, -1 for the root          Joint.m:56-92       */
    const int* type;         /* RMX_JOINT_*                                   JointRevolute/Prismatic/Fixed.m */
    const double* axis;      /* [n][3] unit joint axis                        JointRevolute.m:14   */
    const double* E0_pj;     /* [n][16] joint wrt parent joint at q=0         Joint.setJointTransform :95-99 */
    const double* E0_ji;     /* [n][16] body wrt joint                        Body.setBodyTransform :46-51   */
    const double* I_i;       /* [n][6] diagonal body inertia                  se3.inertiaCuboid :366-379     */
    const double* qRest;     /* rest position of the joint spring (= initial q, Joint.m:157)     */
    const double* tau;       /* joint torque                                  Joint.m:446          */
    const double* stiffness; /* Joint.setStiffness                                                 */
    const double* damping;   /* Joint.setDamping                                                   */
    const double* qLimL;     /* joint limits                                  Joint.m:449-454      */
    const double* qLimU;
    const double* qLimK;
    const double* qLimD;
    double grav[3];          /* Scene.grav                                    Scene.m:48           */
    /* multi-DOF joints (may be NULL when the scene has none) */
    const double* plane;     /* [n][6] JointPlanar: the two in-plane directions b1, b2 (normalised by the caller,
                                JointPlanar.m:16-17); NULL = x-y plane                              */
    const double* qRestR;    /* [nr] rest position of EVERY DOF in reduced order (Joint.m:157 qRest = q); overrides qRest.
                                qRest[n] alone only reaches the first DOF of a multi-DOF joint      */
} rmx_model_desc;

/* Newton constants of driverRedMaxBDF1.m:95-98; rmx_opts_default() fills the reference values. */
typedef struct rmx_opts {
    double h;              /* time step (Scene.h)                                  */
    double tol;            /* 1e-9   ||g||_2 convergence threshold                 */
    double dxMax;          /* 1e3    "Newton diverged" threshold on ||dx||_2       */
    int iterMaxPerDof;     /* 10     iterMax = iterMaxPerDof * nr                  */
    int iterLsMax;         /* 20     line-search halvings                          */
    int lu_mode;           /* dx = -H\g (driverRedMaxBDF1.m:117, LU with partial pivoting in MATLAB):
                              0 (default) eliminate on the diagonal under a growth guard (|multiplier| <= 8 on the symmetrically
                                equilibrated matrix, i.e. threshold pivoting with tau = 1/8; trees of up to 64 nodes also require
                                positive pivots, larger trees - blocked LU, rmx_big.hip big_solve_diag - take pivots of either sign)
                                and redo the solve with full partial pivoting when the guard trips (RMX_ST_PIVOTED);
                              1 always full partial pivoting (the reference behaviour, ~2x slower solve).  The pivot search compares
                                |H(a,k)| as full doubles, lowest row first among equals: LAPACK's first maximum (idamax in dgetf2),
                                in every kernel of the library (ABI 108; up to ABI 107 the one-wavefront kernels compared the top 26
                                bits only); the elimination scales by the reciprocal of the pivot, as dgetf2 does  */
    int compensated;       /* the Newton iterate of newton() (driverRedMaxBDF1.m:94-157):
                              1 (default) carried as an unevaluated sum x + xlo, |xlo| <= ulp(x)/2; xlo enters the residual where x
                                enters linearly with large coefficients (dqtmp = q1 - q0 - h qdot0 and qdot1 = (q1 - q0)/h,
                                :167-169) and nowhere else, and is dropped when the step stores q1.  On long chains in cgs units
                                |M| ulp(q) reaches the reference's tol = 1e-9: on the lattice of doubles ||g|| < tol is then met only
                                at lucky points, which MATLAB's Newton finds because its own evaluation noise dithers the update
                                and a smoother evaluation does not.  With xlo no Newton correction is lost to
                                the rounding of x and the iteration converges to the evaluation noise, at the reference's constants;
                              0 plain doubles: the reference's arithmetic, decision for decision  */
    int ls_fail_limit;     /* straggler policy for batches, 0 (default) = off = the reference: newton() keeps iterating after a line
                              search that ran out its iterLsMax trials without a decrease of ||g|| (:126-138), up to iterMax = 10 nr
                              times - at a non-smooth point of the residual (stick/slip or touch-down of a ground contact) every one of
                              them fails or creeps the same way, ~4600 evaluations for one step of one rollout, and the launch of a whole
                              batch ends with its slowest wavefront.  N > 0: the Newton loop of a step ends ("did not converge", status
                              RMX_ST_MAXITER | RMX_ST_LS_CUT) at the N-th failed line search of that step (failed and barely
                              successful ones alternate at such a point, so they are not required to be consecutive); the iterate
                              is the one that line search left (x0 + 2^-19 dx, as in the reference).  The state after such a step
                              differs from the reference's by the drift of the iterations not run; a step in which no line search
                              fails N times runs exactly as without the option  */
} rmx_opts;

typedef struct rmx_model rmx_model;
typedef struct rmx_batch rmx_batch;

const char* rmx_last_error(void);
int rmx_version(void);
int rmx_device_count(void);
void rmx_opts_default(rmx_opts* o);

/* Scene.init(): validates the listing, orders it depth-first, counts DOFs leaf-to-root and uploads
 * the constant per-joint data.  (Scene.m:59-119, Joint.countDofs :149-158, Body.countDofs :54-60) */
int rmx_model_create(const rmx_model_desc* desc, int device, rmx_model** out);
void rmx_model_destroy(rmx_model* m);
int rmx_model_nr(const rmx_model* m);      /* redmax.Scene.countR()  Scene.m:410-421 */
int rmx_model_nm(const rmx_model* m);      /* redmax.Scene.countM()  Scene.m:398-409 */
/* idx[n]: reduced index of each listed joint, -1 for a fixed joint (Joint.idxR) */
int rmx_model_idxR(const rmx_model* m, int* idx);

/* ForceGroundCuboid (matlab-diff/+redmax/ForceGroundCuboid.m:18-48, 54-183): penalty ground contact with Coulomb friction
 * on the 8 corners of the flagged cuboid bodies -- normal spring kn and damper kd, tangential stick spring kt, friction mu
 * (static/dynamic branch per corner, :112-150), energy 0.5 kn d^2 (:176).  Replaces, for one ground frame per scene,
 *     f = redmax.ForceGroundCuboid(body); f.setTransform(E); f.setStiffness(kn,kt); f.setDamping(kd); f.setFriction(mu);
 * (scenesRedMax.m:303-309).  Arrays follow the scene listing like rmx_model_desc.  Call before stepping; BDF1/BDF2/eval/energy
 * honour it, rmx_step_euler and rmx_adjoint_bdf1 refuse such a model.  All flags zero removes the contact.
 * The reference keeps its force objects in a list (Force.m:26-56) and a body may carry several (a floor and a wall).  This struct
 * holds ONE object per listing entry: a host lists every further force of a body as an extra entry of rmx_model_desc - a
 * RMX_JOINT_FIXED child of the body's joint, E0_pj = identity, E0_ji and sides those of the body, I_i = 0 - flagged here with its own
 * frame and constants.  Same corners, same twist: the same wrench, K and D through the same Jacobian rows; no DOF is added
 * (redmax_amd.Scene.desc() and matlab/+redmax/flattenScene.m do this; tests/test_oracle_fd.py, tests/test_gpu_contact.py). */
typedef struct rmx_ground_contact {
    const int* flags;        /* [n] 1: this body carries a ForceGroundCuboid                       */
    const double* sides;     /* [n][3] cuboid side lengths (BodyCuboid.sides)  ForceGroundCuboid.m:71-75 */
    double E[16];            /* ground frame, column-major 4x4; its Z axis is the plane normal    :30-32, 56-57 */
    double kn, kt;           /* setStiffness(kn, kt)   :35-38 */
    double mu;               /* setFriction(mu)        :46-48 */
    double kd;               /* setDamping(kd)         :41-43 */
    /* Every ForceGroundCuboid object holds its own E, kn, kt, mu, kd (:6-13): scenes whose force objects differ (a floor and a wall,
     * a slippery patch) pass them per body, in listing order; NULL = the shared value above for every flagged body (ABI 106). */
    const double* E_body;    /* [n][16] column-major 4x4 per body, or NULL */
    const double* kn_body;   /* [n] or NULL */
    const double* kt_body;   /* [n] or NULL */
    const double* mu_body;   /* [n] or NULL */
    const double* kd_body;   /* [n] or NULL */
} rmx_ground_contact;
int rmx_model_set_ground_contact(rmx_model* m, const rmx_ground_contact* gc);

/* JointSpherical / JointFree3D (JointSpherical.m:4-17, 28-34, 63-102): every such joint is in one of 12 Euler charts, numbered as
 * the reference's CHART_* constants (1 XYX, 2 XZX, 3 YZY, 4 YXY, 5 ZXZ, 6 ZYZ, 7 XYZ, 8 XZY, 9 YZX, 10 YXZ, 11 ZXY, 12 ZYX) and
 * constructed in CHART_XYZ.  rmx_step_bdf1/bdf2 run reparam_ after every step per trajectory (status bit RMX_ST_CHART when a
 * chart changed), so q/qdot returned by rmx_get_state are coordinates in the charts rmx_get_charts reports.  rmx_set_state
 * puts every joint back to CHART_XYZ; rmx_set_charts (after it) declares other charts for the given coordinates.
 * charts: host [batch][nsph], spherical joints in listing order.  rmx_step_euler / rmx_adjoint_bdf1 refuse such models.  A scene may
 * hold as many JointSpherical / JointFree3D joints as its node limit allows (256 nodes: 85; ABI 109 - 21 up to ABI 108). */
int rmx_model_nsph(const rmx_model* m);
int rmx_get_charts(rmx_batch* b, int* charts);
int rmx_set_charts(rmx_batch* b, const int* charts);

/* `batch` independent trajectories of the model, state resident in HBM on the model's device. */
int rmx_batch_create(rmx_model* m, int batch, rmx_batch** out);
void rmx_batch_destroy(rmx_batch* b);
int rmx_batch_size(const rmx_batch* b);

/* Joint.setQ / Joint.getQ (Joint.m:173-292) for the whole batch: host arrays [batch][nr]. */
int rmx_set_state(rmx_batch* b, const double* q, const double* qdot);
int rmx_get_state(rmx_batch* b, double* q, double* qdot);
/* Same, with DEVICE pointers (no host round trip; used to feed the final RCCL gather). */
int rmx_set_state_device(rmx_batch* b, const double* d_q, const double* d_qdot);
int rmx_get_state_device(rmx_batch* b, double* d_q, double* d_qdot);

/* Parity hook = evalBDF1 / evalSDIRK2a / evalSDIRK2b / evalBDF2 (driverRedMaxBDF1.m:160-187,
 * driverRedMaxBDF2.m:194-293) in their common form
 *     qdot = (q - qA)/eta ;  dqtmp = q - qB ;  g = M dqtmp - eta^2 f ;  H = M - eta D - eta^2 K + dMdq dqtmp
 * evaluated for every trajectory.  BDF1: eta = h, qA = q0, qB = q0 + h qdot0.
 * q,qA,qB: host [batch][nr].  g: host [batch][nr].  H: host [batch][nr*nr] column-major or NULL
 * (NULL selects the cheap residual-only path, nargout==1 in the reference).  Does not change state. */
int rmx_eval(rmx_batch* b, const double* q, const double* qA, const double* qB, double eta,
             double* g, double* H);

/* Parity hook = computeValues (driverRedMaxBDF1.m:190-243) at (q, qdot) for every trajectory: the reduced mass matrix
 * M = J'MmJ (:212), the force vector f = fr + J'(fm - Mm Jdot qdot) (:215-216) and D = df/dqdot (:227-237).  q, qdot, f: host
 * [batch][nr]; M, D: host [batch][nr*nr] column-major.  K = df/dq is not returned on its own: it only exists folded into H
 * (rmx_eval gives H = M - eta D - eta^2 K + dMdq dqtmp).  Spherical joints: in the coordinates of the batch's current Euler charts
 * (rmx_get_charts).  ForceGroundCuboid: its wrench is part of f, its damping block J' Dm J part of D (ForceGroundCuboid.m:104-106).
 * Does not change state. */
int rmx_eval_mfd(rmx_batch* b, const double* q, const double* qdot, double* M, double* f, double* D);

/* computeValues' FULL output, [M, f, dMdq, K, D] (driverRedMaxBDF1.m:188-243), at (q, qdot) for every trajectory, for trees of ANY
 * size the library takes (ABI 108).  Every output may be NULL.
 *   M, D, K : [batch][nr*nr] column-major;  K = df/dq (:239-243: Kr + J'KmJ + Kqvv + the dJdq terms), D = df/dqdot, M = J'MmJ
 *   f       : [batch][nr]
 *   dMv     : [batch][nr*nr] column-major, column i = dMdq(:,:,i) v for the caller's v [batch][nr] - the form in which evalBDF1
 *             consumes the tensor (:181-184, v = dqtmp); needs v
 *   dMdq    : the tensor itself, [batch][nr*nr*nr], entry (r, c, i) at r + nr (c + nr i)  (nr evaluations: a test hook)
 * The world-frame kernels never form K, D or the tensor on their own: H(eta; v) = M + dMdq v - eta D - eta^2 K is what
 * they evaluate, exactly, for any eta and v.  This hook takes the pieces apart on the host from a few such evaluations at the SAME
 * (q, qdot) (qA = q - eta qdot): v = 0 at eta = 1 (and 2, 1/2 for trees of more than 64 nodes, where M and D have no kernel of their
 * own) gives M, D, K; one more with the caller's v gives dMv = H(1; v) - H(1; 0).
 * Accuracy: up to 64 nodes M and D come from their own kernel and K = (M - D) - H(1; 0) carries an absolute error of order eps |M|.  Larger
 * trees: an evaluation of H(eta) carries roundoff of order eps (|M| + eta |D| + eta^2 |K|), so each piece is taken from a triple H(e),
 * H(2 e), H(e / 2) at its own power-of-two e - K at e_K ~ sqrt(|M| / |K|), D at min(|M| / |D|, e_K), M at the smallest of these and 1 - which
 * leaves every piece with an error of order eps times ITS OWN norm (plus the cross terms eps |D| sqrt(|K| / |M|) in K and
 * eps sqrt(|M| |K|) in D); 3 to 9 evaluations (ABI 109; ABI 108 took all three at e = 1: eps |M| absolute in D and K).  Same conventions as rmx_eval_mfd for Euler
 * charts and ForceGroundCuboid.  Does not change state. */
int rmx_compute_values(rmx_batch* b, const double* q, const double* qdot, const double* v,
                       double* M, double* f, double* D, double* K, double* dMv, double* dMdq);

/* Per-trajectory counters of one rmx_step_* call (host arrays [batch], any may be NULL). */
typedef struct rmx_stats {
    int* newton_iters;   /* Newton iterations summed over the steps                    */
    int* ls_halvings;    /* line-search halvings summed over the steps                 */
    int* status;         /* OR of RMX_ST_* over the steps                              */
} rmx_stats;

/* simLoop of driverRedMaxBDF1.m:57-91: nsteps fully-implicit BDF1 steps for every trajectory in ONE
 * kernel launch (trajectories are independent, so the step loop runs on the device).
 * hist_T/hist_V: optional host [nsteps][batch] kinetic / potential energy after each step
 * (Scene.saveHistory, Scene.m:134-161; Joint/Body.computeEnergies).  May be NULL. */
int rmx_step_bdf1(rmx_batch* b, const rmx_opts* opts, int nsteps, rmx_stats* stats,
                  double* hist_T, double* hist_V);

/* The same two loops with the full per-step record of Scene.saveHistory (Scene.m:134-161: q, qdot, T, V after every step;
 * t = k*h).  integrator: 1 = BDF1 (driverRedMaxBDF1.m), 2 = BDF2 (driverRedMaxBDF2.m), the reference's itype numbering
 * (scenesRedMax.m:5-6).  Host arrays, any pair may be NULL. */
typedef struct rmx_history {
    double* T;       /* [nsteps][batch]      kinetic energy   (with V)      */
    double* V;       /* [nsteps][batch]      potential energy                */
    double* q;       /* [nsteps][batch][nr]  reduced positions (with qdot)   */
    double* qdot;    /* [nsteps][batch][nr]  reduced velocities              */
    int* charts;     /* [nsteps][batch][nsph] Euler chart (1..12) of every JointSpherical / JointFree3D after each step, i.e. the
                        chart the step's q / qdot are expressed in (the reference keeps chart and q together on the joint,
                        JointSpherical.m:28-34; its own history has a TODO for it, Scene.m:138).  May be NULL; ignored when the
                        model has no spherical joint */
} rmx_history;
int rmx_step_history(rmx_batch* b, const rmx_opts* opts, int nsteps, int integrator, rmx_stats* stats,
                     const rmx_history* hist);

/* simLoop of driverRedMaxBDF2.m:57-125: the first call after set_state takes the SDIRK2 start step
 * (two Newton solves, :64-88), later steps are BDF2 (:89-106).  The batch keeps (q,qdot) of step k-1. */
int rmx_step_bdf2(rmx_batch* b, const rmx_opts* opts, int nsteps, rmx_stats* stats,
                  double* hist_T, double* hist_V);

/* TaskBDF1PointPos (matlab-diff/+redmax/TaskBDF1PointPos.m): bring a point of a body to a world target at one step;
 * the parameters are constant joint torques tau = pscale * p. */
typedef struct rmx_task_pointpos {
    int body;              /* listing index of the body             setBody                       */
    double xlocal[3];      /* point in body coordinates             setPoint                      */
    double xtarget[3];     /* world target                          setTarget                     */
    int step;              /* 1-based step at which the point is measured (setTime: step = round(t/h)) */
    double pscale;         /* torque scale                          setScale                      */
    double wreg, wpos;     /* regulariser / position weights        setWeights                    */
} rmx_task_pointpos;

/* taskObjective of driverRedMaxAdjointBDF1.m:39-62, batched (BASELINE.json configs[3]): starting from the batch's current
 * state, forward simLoop (:65-102) with the line-search-free newton (:105-146; opts->iterMaxPerDof should be 5 as at :108)
 * under the torques tau = pscale*p, storing H, M, D of the last evaluated iterate of every step in HBM, then the backward
 * sweep TaskBDF1.calcFinal (TaskBDF1.m:45-81).  p: host [batch][nr]; P: host [batch]; dPdp: host [batch][nr].
 * The batch state is left at the end of the forward rollout.  The forward solves take diagonal pivots under the growth guard
 * first, as the step kernels do (RMX_ST_PIVOTED in stats->status when one was redone with partial pivoting; the reference's
 * lu(H,'vector') always pivots, :127). */
int rmx_adjoint_bdf1(rmx_batch* b, const rmx_opts* opts, int nsteps, const rmx_task_pointpos* task, const double* p,
                     double* P, double* dPdp, rmx_stats* stats);

/* The same for driverRedMaxAdjointBDF2.m:38-62 with TaskBDF2 / TaskBDF2PointPos (matlab-diff/+redmax/TaskBDF2.m, TaskBDF2PointPos.m;
 * scene 101, scenesRedMax.m:437-471): the forward simLoop (:65-136) takes the SDIRK2a / SDIRK2b start step and BDF2 steps with the
 * line-search-free newton (:139-181), the backward sweep is TaskBDF2.calcFinal (TaskBDF2.m:45-107: four off-diagonal blocks per
 * step, the k == 1 variants with the SDIRK2 coefficients).  As in the reference dg/dp = -(4/9) h^2 pscale I is used for every step
 * and dg/dqa of the start step is dropped (TaskBDF2PointPos.m:97-106, TaskBDF2.m:52-55), so dPdp carries the reference's own
 * O(1/nsteps) start-step error.  Arguments as rmx_adjoint_bdf1.  The batch is left at the end of the forward rollout with the
 * BDF2 history in place (rmx_step_bdf2 may continue it). */
int rmx_adjoint_bdf2(rmx_batch* b, const rmx_opts* opts, int nsteps, const rmx_task_pointpos* task, const double* p,
                     double* P, double* dPdp, rmx_stats* stats);

/* The same two calls with DEVICE pointers for p, P and dPdp (ABI 110): an optimiser that lives on the device - or a caller that
 * evaluates many parameter batches - pays no host round trip for them (three small copies and their synchronisation were a sixth
 * of the 20-step configs[3] call).  d_p: [batch][nr], d_P: [batch], d_dPdp: [batch][nr], none of them retained; stats (host
 * arrays, may be NULL) as above.  The call returns when the kernels have finished. */
int rmx_adjoint_bdf1_device(rmx_batch* b, const rmx_opts* opts, int nsteps, const rmx_task_pointpos* task, const double* d_p,
                            double* d_P, double* d_dPdp, rmx_stats* stats);
int rmx_adjoint_bdf2_device(rmx_batch* b, const rmx_opts* opts, int nsteps, const rmx_task_pointpos* task, const double* d_p,
                            double* d_P, double* d_dPdp, rmx_stats* stats);

/* euler() of matlab-simple/testRedMax.m:67-109 (BASELINE.json configs[0]): nsteps linearly-implicit Euler steps,
 *   Mr = J'MmJ ; (Mr + h Dr - h^2 Kr) qdot1 = Mr qdot0 + h (J'(fm - Mm Jdot qdot0) + fr) ; q1 = q0 + h qdot1.
 * hist_T/hist_V as in rmx_step_bdf1. */
int rmx_step_euler(rmx_batch* b, double h, int nsteps, double* hist_T, double* hist_V);

/* Joint.computeEnergies + Body.computeEnergies at the current state (Joint.m:616-637, Body.m:167-173):
 * host arrays [batch]. */
int rmx_energy(rmx_batch* b, double* T, double* V);

/* The HIP stream the batch's kernels are enqueued on (as void*), for callers that order other work
 * (e.g. a torch.distributed gather) after it. */
void* rmx_batch_stream(const rmx_batch* b);
/* ---- Asynchronous stepping: one host thread (MATLAB has one) driving several batches / devices at once (ABI 107).
 * simLoop (driverRedMaxBDF1.m:57-91, driverRedMaxBDF2.m:57-125) of a batch is ONE kernel launch on the batch's own stream; the
 * *_async entries enqueue it and return without waiting, so a host loop over batches that live on different devices (or on one
 * device) has all of them running before it blocks on the first rmx_sync().  Rules: between an *_async call and rmx_sync() on the
 * same batch only *_async / rmx_sync / rmx_stats_reset may be called on it; other batches are free.  The per-trajectory counters
 * accumulate on the device (rmx_stats_reset before, rmx_stats_read after the sync).
 *   rmx_step_bdf1_async / rmx_step_bdf2_async   nsteps steps, no per-step record
 *   rmx_step_history_async                      integrator 1 | 2 as rmx_step_history; `record` chooses what Scene.saveHistory
 *                                               (Scene.m:134-161) keeps ON THE DEVICE for every step: RMX_REC_ENERGY (T, V),
 *                                               RMX_REC_STATE (q, qdot), RMX_REC_CHARTS (Euler charts), OR-ed together
 *   rmx_sync                                    waits for the batch's stream; reports a failed launch
 *   rmx_history_read                            after rmx_sync: copies the record of the last rmx_step_history_async into host arrays
 *                                               shaped as in rmx_history (any pointer whose part was not recorded must be NULL);
 *                                               the record stays readable until the next step call on the batch */
enum { RMX_REC_ENERGY = 1, RMX_REC_STATE = 2, RMX_REC_CHARTS = 4 };
int rmx_step_bdf1_async(rmx_batch* b, const rmx_opts* opts, int nsteps);
int rmx_step_bdf2_async(rmx_batch* b, const rmx_opts* opts, int nsteps);
int rmx_step_history_async(rmx_batch* b, const rmx_opts* opts, int nsteps, int integrator, int record);
int rmx_sync(rmx_batch* b);
int rmx_history_read(rmx_batch* b, const rmx_history* hist);

/* ---- Multi-device groups (ABI 107): BASELINE.json's north_star shards the batch axis across GPUs and names MATLAB as the host.
 * A group owns one model + one batch per listed device (a device may be listed more than once: its shards then share it, each on
 * its own stream) and splits `batch` trajectories into contiguous shards whose sizes differ by at most one, in device-list order
 * (the "strong" plan of redmax_amd/sharding.py).  Every array argument is the WHOLE batch, shaped exactly as for a single rmx_batch;
 * the group scatters / gathers the shards' slices.  rmx_group_step is simLoop for the whole batch: it launches every shard's
 * kernel asynchronously, then waits for all of them and gathers the counters and the per-step record into the caller's arrays -
 * the one "collective" of the path, here a set of device-to-host copies (no RCCL: the host array is the destination).
 * gc may be NULL (no ForceGroundCuboid).  devices == NULL: devices 0 .. ndevices-1. */
typedef struct rmx_group rmx_group;
int rmx_group_create(const rmx_model_desc* desc, const rmx_ground_contact* gc, int batch, const int* devices, int ndevices,
                     rmx_group** out);
void rmx_group_destroy(rmx_group* g);
int rmx_group_batch_size(const rmx_group* g);
int rmx_group_nshards(const rmx_group* g);
/* shard s: its device, the index of its first trajectory in the whole batch and its trajectory count (any pointer may be NULL) */
int rmx_group_shard(const rmx_group* g, int s, int* device, int* first, int* count);
/* the shard's own batch / model, for the hooks that have no group form (rmx_eval, rmx_eval_mfd, rmx_adjoint_*, rmx_step_euler,
 * rmx_get_charts ...): call them per shard with the array pointers advanced to the shard's first trajectory */
rmx_batch* rmx_group_shard_batch(rmx_group* g, int s);
rmx_model* rmx_group_shard_model(rmx_group* g, int s);
int rmx_group_set_state(rmx_group* g, const double* q, const double* qdot);     /* [batch][nr], as rmx_set_state */
int rmx_group_get_state(rmx_group* g, double* q, double* qdot);
/* stats, hist: as rmx_step_history, arrays for the whole batch; hist (and any of its members) may be NULL */
int rmx_group_step(rmx_group* g, const rmx_opts* opts, int nsteps, int integrator, rmx_stats* stats, const rmx_history* hist);
/* the two halves of rmx_group_step for hosts that overlap their own work with the devices: launch all shards (record: RMX_REC_*),
 * then wait and gather (stats / hist as above; hist members outside `record` must be NULL) */
int rmx_group_step_async(rmx_group* g, const rmx_opts* opts, int nsteps, int integrator, int record);
int rmx_group_sync(rmx_group* g, rmx_stats* stats, const rmx_history* hist);
int rmx_group_energy(rmx_group* g, double* T, double* V);                      /* [batch], as rmx_energy */
/* The final gather with DEVICE-resident destinations (ABI 111) - BASELINE.json north_star: "the batch axis shards naturally across GPUs
 * with RCCL over xGMI only for the final gather"; SURVEY.md 8(e): one collective at the end of the rollout, [B/N][nr] x 2 per shard.
 * d_q[s], d_qdot[s] (s < nshards): device pointers to [batch][nr] arrays ON SHARD s's DEVICE (a shard that receives nothing may pass
 * NULL).  root_or_all = RMX_GATHER_ALL: every shard's device receives the whole batch; a shard index: that shard's device alone.
 * Groups whose devices are pairwise distinct form a single-process RCCL clique at the first call (ncclCommInitAll on the device list;
 * librccl is bound at run time, the library itself links the HIP runtime only) and run ncclAllGather (equal shards), one
 * ncclBroadcast per shard inside a group call (shards that differ by one rollout) or ncclSend / ncclRecv (one root), enqueued on
 * the shards' own streams behind their step kernels.  A group that lists a device more than once - RCCL refuses such a clique -
 * exchanges by device-to-device / peer copies on the same streams (RMX_GROUP_GATHER=copy in the environment forces that path).
 * Returns when the destinations are complete.  rmx_group_gather_path: how the last gather travelled ("rccl:allgather",
 * "rccl:broadcast", "rccl:sendrecv", "copy"; "" before the first). */
enum { RMX_GATHER_ALL = -1 };
int rmx_group_gather_device(rmx_group* g, double* const* d_q, double* const* d_qdot, int root_or_all);
const char* rmx_group_gather_path(const rmx_group* g);
/* The same gather into [batch][nr] destinations the GROUP owns on every receiving shard's device (allocated at the first call; for
 * hosts that cannot allocate device memory themselves - the MEX gateway's 'gather').  rmx_group_gathered: the device pointers of shard
 * s's copy (valid until the group is destroyed; what a MATLAB host would wrap as gpuArray); rmx_group_gathered_read: that copy into host
 * arrays ([batch][nr]; either pointer may be NULL). */
int rmx_group_gather(rmx_group* g, int root_or_all);
int rmx_group_gathered(rmx_group* g, int s, const double** d_q, const double** d_qdot);
int rmx_group_gathered_read(rmx_group* g, int s, double* q, double* qdot);
/* Timing of the last rmx_group_step / rmx_group_step_async + rmx_group_sync: wall_ms = host wall clock from the first launch to
 * the last shard's completion; per shard (arrays [nshards], any may be NULL) kernel_ms = HIP-event time of its launch, start_ms /
 * end_ms = when its launch began / ended relative to the start of the FIRST shard on the same device (HIP events of one device
 * share a clock; 0 / kernel_ms for that first shard).  Shards run concurrently when start_ms of one lies before end_ms of another. */
int rmx_group_timing(rmx_group* g, double* wall_ms, double* kernel_ms, double* start_ms, double* end_ms);
/* Per-trajectory counters kept on the device across *_async calls (see "Asynchronous stepping" above). */
int rmx_stats_reset(rmx_batch* b);
int rmx_stats_read(rmx_batch* b, rmx_stats* stats);

#ifdef __cplusplus
}
#endif
#endif
