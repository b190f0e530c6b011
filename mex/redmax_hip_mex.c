/*
 * redmax_hip_mex.c -- MATLAB MEX gateway onto the C ABI of include/redmax_hip.h.
 *
 * The reference (sueda/redmax, matlab-diff) runs simLoop / newton / evalBDF1 / computeValues inside the MATLAB
 * interpreter (driverRedMaxBDF1.m:57-243).  This gateway is the thin layer BASELINE.json's north_star names: MATLAB keeps the
 * +redmax Scene/Joint/Body classes and the driverRedMaxBDF1(sceneID,batch) entry point, and the body of simLoop becomes
 * calls into libredmax_hip.so.  The MATLAB callers are matlab/+redmax/HipSim.m (handle wrapper), matlab/+redmax/flattenScene.m
 * (Scene -> rmx_model_desc arrays) and matlab/driverRedMaxBDF1.m / driverRedMaxBDF2.m.
 *
 *   mex -I../include redmax_hip_mex.c -L../redmax_amd -lredmax_hip          (inside MATLAB, R2018a+ for the C matrix API used)
 *
 * MATLAB is not available in the build environment: the file is compiled and EXECUTED against mex/stub/ (a minimal
 * implementation of the mx and mex functions used here) by tests/test_mex_gateway.py.
 *
 * Commands (first argument is the command string).  Array shapes are MATLAB's; the ABI's row-major [batch][nr] is the
 * column-major nr x batch MATLAB matrix, [nsteps][batch] is batch x nsteps, [nsteps][batch][nr] is nr x batch x nsteps, 4x4
 * transforms are column-major on both sides, so no transposition happens anywhere.
 *
 *   v            = redmax_hip_mex('version')
 *   n            = redmax_hip_mex('devices')
 *   h            = redmax_hip_mex('create', desc, batch [, devices])    desc: struct, see cmd_create() below.  devices: a device index
 *                  or a VECTOR of them (default 0).  The batch is split into one contiguous shard per listed device (sizes differ by at
 *                  most one; a device may be listed twice); every array below is the WHOLE batch, the gateway scatters / gathers
 *                  (rmx_group_*).  BASELINE.json north_star: "the batch axis shards across GPUs", host = MATLAB.
 *                  redmax_hip_mex('destroy', h)
 *   info         = redmax_hip_mex('info', h)                            struct nr, nm, nsph, batch, idxR (0-based, -1 fixed),
 *                                                                       nshards, devices, shard_first (0-based), shard_count
 *                  redmax_hip_mex('set', h, q, qdot)                     nr x B each          Joint.setQ   (Joint.m:231-292)
 *   [q, qdot]    = redmax_hip_mex('get', h)                                                   Joint.getQ   (Joint.m:173-229)
 *   [T,V,st,Q,Qd,C]= redmax_hip_mex('step', h, itype, hstep, nsteps [, opts])
 *                  itype 1: simLoop of driverRedMaxBDF1.m:57-91, 2: of driverRedMaxBDF2.m:57-125.  T, V: B x nsteps
 *                  (Scene.saveHistory energies); st: B x 3 int32 [newton iterations, line-search halvings, RMX_ST_* bits];
 *                  Q, Qd (only when requested): nr x B x nsteps, the full Scene.saveHistory record (Scene.m:134-161);
 *                  C (only when requested): nsph x B x nsteps int32, the Euler chart of every spherical joint after each step.
 *                  opts: struct with any of tol, dxMax, iterMaxPerDof, iterLsMax, lu_mode, compensated, ls_fail_limit (driverRedMaxBDF1.m:95-98).
 *                  All shards' kernels are launched before the first one is waited for: N devices run concurrently.
 *                  redmax_hip_mex('step_async', h, itype, hstep, nsteps [, opts [, record]])
 *                  the same launch without the wait: MATLAB gets control back while the devices step (several handles can be in
 *                  flight).  record: bit mask of what Scene.saveHistory keeps on the device, 1 = T, V (default), 2 = Q, Qdot, 4 = C.
 *   [T,V,st,Q,Qd,C]= redmax_hip_mex('sync', h)
 *                  waits for the launches of the last 'step_async' and gathers; outputs beyond what `record` asked for are an error.
 *   [wall,k,t0,t1] = redmax_hip_mex('timing', h)                         rmx_group_timing of the last step: wall clock ms, and per shard
 *                  (1 x nshards) kernel ms, start and end of its launch relative to the first shard on the same device
 *   [T, V]       = redmax_hip_mex('euler', h, hstep, nsteps)             matlab-simple/testRedMax.m:67-109
 *   [g, H]       = redmax_hip_mex('eval', h, q, qA, qB, eta)             evalBDF1 & co (driverRedMaxBDF1.m:160-187); H: nr x nr x B
 *   [T, V]       = redmax_hip_mex('energy', h)                           Joint/Body.computeEnergies
 *   c            = redmax_hip_mex('getcharts', h)                        nsph x B int32, JointSpherical.chart
 *                  redmax_hip_mex('setcharts', h, c)
 *   [P,dPdp,st]  = redmax_hip_mex('adjoint', h, hstep, nsteps, task, p [, integrator])  taskObjective of
 *                  driverRedMaxAdjointBDF1.m:39-62 (integrator 1, default) or driverRedMaxAdjointBDF2.m:38-62 (integrator 2);
 *                  task: struct body (1-based listing index), xlocal, xtarget, step, pscale, wreg, wpos; p: nr x B;
 *                  st: B x 2 int32 [newton iterations, status]
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "mex.h"
#include "redmax_hip.h"
#include "redmax_hip_profile.h"   /* 'timing' / 'ticks': measurement hooks, not part of the host-facing ABI */

typedef struct {
    uint64_t magic;
    rmx_group* g;          /* one model + batch per listed device */
    int nshards;
    int nr, nm, nsph, B, njoints;
    int pending_nsteps;    /* 'step_async' in flight: its nsteps and record mask ('sync' shapes its outputs from them) */
    int pending_record;
    int pending;
} handle_t;

#define HANDLE_MAGIC 0x726d78686970ull /* "rmxhip" */
/* live handles: a handle value that MATLAB passes back is only dereferenced when it is in this table, so a stale or
 * made-up uint64 is an error message, not a crash */
#define MAX_LIVE 256
#define MAX_SHARDS 64
static handle_t* g_live[MAX_LIVE];
static int live_slot(const handle_t* h) {
    for (int i = 0; i < MAX_LIVE; ++i)
        if (g_live[i] == h) return i;
    return -1;
}

static void die_rmx(const char* what) { mexErrMsgIdAndTxt("redmax:hip", "%s: %s", what, rmx_last_error()); }
static void die(const char* msg) { mexErrMsgIdAndTxt("redmax:hip", "%s", msg); }

static const mxArray* field(const mxArray* s, const char* k, int required) {
    const mxArray* f = mxIsStruct(s) ? mxGetField(s, 0, k) : NULL;
    if ((!f || mxIsEmpty(f)) && required) mexErrMsgIdAndTxt("redmax:hip", "desc.%s is required", k);
    return (f && !mxIsEmpty(f)) ? f : NULL;
}
/* double array field with exactly `count` elements (NULL when optional and absent) */
static const double* f64(const mxArray* s, const char* k, size_t count, int required) {
    const mxArray* f = field(s, k, required);
    if (!f) return NULL;
    if (!mxIsDouble(f) || mxIsComplex(f)) mexErrMsgIdAndTxt("redmax:hip", "desc.%s must be a real double array", k);
    if (mxGetNumberOfElements(f) != count) mexErrMsgIdAndTxt("redmax:hip", "desc.%s must have %d elements", k, (int)count);
    return mxGetPr(f);
}
static const int* i32(const mxArray* s, const char* k, size_t count, int required) {
    const mxArray* f = field(s, k, required);
    if (!f) return NULL;
    if (!mxIsInt32(f)) mexErrMsgIdAndTxt("redmax:hip", "desc.%s must be an int32 array", k);
    if (mxGetNumberOfElements(f) != count) mexErrMsgIdAndTxt("redmax:hip", "desc.%s must have %d elements", k, (int)count);
    return (const int*)mxGetData(f);
}
static double scalar_field(const mxArray* s, const char* k, double dflt) {
    const mxArray* f = field(s, k, 0);
    return f ? mxGetScalar(f) : dflt;
}

static handle_t* get_handle(int nrhs, const mxArray* prhs[]) {
    if (nrhs < 2 || !mxIsUint64(prhs[1]) || mxGetNumberOfElements(prhs[1]) != 1) die("second argument must be the uint64 handle");
    handle_t* h = (handle_t*)(uintptr_t)(*(const uint64_t*)mxGetData(prhs[1]));
    if (!h || live_slot(h) < 0 || h->magic != HANDLE_MAGIC) die("stale or invalid handle");
    return h;
}
static const double* state_arg(const mxArray* a, const handle_t* h, const char* name) {
    if (!mxIsDouble(a) || mxIsComplex(a) || mxGetNumberOfElements(a) != (size_t)h->nr * (size_t)h->B)
        mexErrMsgIdAndTxt("redmax:hip", "%s must be a real double nr x batch (%d x %d) array", name, h->nr, h->B);
    return mxGetPr(a);
}
static mxArray* new_i32(size_t r, size_t c) { return mxCreateNumericMatrix(r, c, mxINT32_CLASS, mxREAL); }
/* the hooks without a group form run shard by shard, every array advanced to the shard's first trajectory */
static rmx_batch* shard(const handle_t* h, int s, size_t* first) {
    int f = 0;
    if (rmx_group_shard(h->g, s, NULL, &f, NULL)) die_rmx("rmx_group_shard");
    *first = (size_t)f;
    return rmx_group_shard_batch(h->g, s);
}

/* Scene.init() (Scene.m:59-119) -> rmx_model_create.  desc fields, joints in the scene's listing order (N = njoints):
 *   njoints; parent int32 1xN (0-based, -1 root); type int32 1xN (RMX_JOINT_*); axis 3xN; E0_pj, E0_ji 4x4xN; I_i 6xN;
 *   qRest, tau, stiffness, damping, qLimL, qLimU, qLimK, qLimD 1xN (optional: the Joint.m:77-83 defaults apply); grav 3x1;
 *   plane 6xN ([b1;b2] of JointPlanar, optional); qRestR nr x 1 (rest position of every DOF, optional);
 *   ground contact (optional, ForceGroundCuboid): contact int32 1xN, sides 3xN, groundE 4x4, kn, kt, mu, kd */
static void cmd_create(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    (void)nlhs;
    if (nrhs < 3) die("usage: h = redmax_hip_mex('create', desc, batch [, devices])");
    const mxArray* s = prhs[1];
    if (!mxIsStruct(s)) die("desc must be a struct (redmax.flattenScene)");
    const int batch = (int)mxGetScalar(prhs[2]);
    int devices[MAX_SHARDS] = {0};
    int ndev = 1;
    if (nrhs > 3 && !mxIsEmpty(prhs[3])) {     /* a device index or a vector of them: one shard per entry */
        if (!mxIsDouble(prhs[3]) || mxIsComplex(prhs[3])) die("devices must be a real double scalar or vector");
        const size_t nd = mxGetNumberOfElements(prhs[3]);
        if (nd > MAX_SHARDS) die("too many devices listed");
        ndev = (int)nd;
        for (int i = 0; i < ndev; ++i) devices[i] = (int)mxGetPr(prhs[3])[i];
    }
    rmx_model_desc d;
    memset(&d, 0, sizeof d);
    d.njoints = (int)scalar_field(s, "njoints", 0);
    if (d.njoints < 1) die("desc.njoints must be >= 1");
    const size_t n = (size_t)d.njoints;
    d.parent = i32(s, "parent", n, 1);
    d.type = i32(s, "type", n, 1);
    d.axis = f64(s, "axis", 3 * n, 1);
    d.E0_pj = f64(s, "E0_pj", 16 * n, 1);
    d.E0_ji = f64(s, "E0_ji", 16 * n, 1);
    d.I_i = f64(s, "I_i", 6 * n, 1);
    d.qRest = f64(s, "qRest", n, 0);
    d.tau = f64(s, "tau", n, 0);
    d.stiffness = f64(s, "stiffness", n, 0);
    d.damping = f64(s, "damping", n, 0);
    d.qLimL = f64(s, "qLimL", n, 0);
    d.qLimU = f64(s, "qLimU", n, 0);
    d.qLimK = f64(s, "qLimK", n, 0);
    d.qLimD = f64(s, "qLimD", n, 0);
    d.plane = f64(s, "plane", 6 * n, 0);
    memcpy(d.grav, f64(s, "grav", 3, 1), 3 * sizeof(double));
    {   /* qRestR has nr entries; nr is only known after lowering, so its length is checked against the DOF counts here */
        const mxArray* f = field(s, "qRestR", 0);
        if (f) {
            static const int ndof_of[] = {0, 1, 1, 2, 3, 2, 3, 3, 6};
            size_t nr = 0;
            for (size_t i = 0; i < n; ++i) {
                if (d.type[i] < 0 || d.type[i] > RMX_JOINT_FREE3D) die("desc.type holds an unknown joint type");
                nr += (size_t)ndof_of[d.type[i]];
            }
            d.qRestR = f64(s, "qRestR", nr, 1);
        }
    }
    /* every field of desc is read and validated BEFORE anything is created: a validation failure leaves through
     * mexErrMsgIdAndTxt (a longjmp out of the MEX function), which must not strand a device model or a persistent handle */
    const int has_contact = field(s, "contact", 0) != NULL;
    rmx_ground_contact gc;
    memset(&gc, 0, sizeof gc);
    if (has_contact) {   /* scene.forces holds ForceGroundCuboid objects (scenesRedMax.m:303-309) */
        gc.flags = i32(s, "contact", n, 1);
        gc.sides = f64(s, "sides", 3 * n, 1);
        memcpy(gc.E, f64(s, "groundE", 16, 1), 16 * sizeof(double));
        gc.kn = scalar_field(s, "kn", 1.0);   /* ForceGroundCuboid.m:22-26 defaults */
        gc.kt = scalar_field(s, "kt", 0.0);
        gc.mu = scalar_field(s, "mu", 0.0);
        gc.kd = scalar_field(s, "kd", 0.0);
        /* one frame / parameter set per force object (ForceGroundCuboid.m:6-13); absent: the shared values above */
        if (field(s, "groundE_body", 0)) gc.E_body = f64(s, "groundE_body", 16 * n, 1);
        if (field(s, "kn_body", 0)) gc.kn_body = f64(s, "kn_body", n, 1);
        if (field(s, "kt_body", 0)) gc.kt_body = f64(s, "kt_body", n, 1);
        if (field(s, "mu_body", 0)) gc.mu_body = f64(s, "mu_body", n, 1);
        if (field(s, "kd_body", 0)) gc.kd_body = f64(s, "kd_body", n, 1);
    }
    const int slot = live_slot(NULL);
    if (slot < 0) die("too many live handles (destroy some first)");
    handle_t* h = (handle_t*)mxCalloc(1, sizeof *h);
    mexMakeMemoryPersistent(h);
    if (rmx_group_create(&d, has_contact ? &gc : NULL, batch, devices, ndev, &h->g)) { mxFree(h); die_rmx("rmx_group_create"); }
    h->nshards = rmx_group_nshards(h->g);
    h->nr = rmx_model_nr(rmx_group_shard_model(h->g, 0));
    h->nm = rmx_model_nm(rmx_group_shard_model(h->g, 0));
    h->nsph = rmx_model_nsph(rmx_group_shard_model(h->g, 0));
    h->B = batch;
    h->njoints = d.njoints;
    h->magic = HANDLE_MAGIC;
    g_live[slot] = h;
    plhs[0] = mxCreateNumericMatrix(1, 1, mxUINT64_CLASS, mxREAL);
    *(uint64_t*)mxGetData(plhs[0]) = (uint64_t)(uintptr_t)h;
}

static void cmd_destroy(int nrhs, const mxArray* prhs[]) {
    handle_t* h = get_handle(nrhs, prhs);
    g_live[live_slot(h)] = NULL;
    rmx_group_destroy(h->g);
    h->magic = 0;
    mxFree(h);
}

static void cmd_info(mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    handle_t* h = get_handle(nrhs, prhs);
    static const char* names[] = {"nr", "nm", "nsph", "batch", "idxR", "nshards", "devices", "shard_first", "shard_count"};
    plhs[0] = mxCreateStructMatrix(1, 1, 9, names);
    mxSetField(plhs[0], 0, "nr", mxCreateDoubleScalar(h->nr));
    mxSetField(plhs[0], 0, "nm", mxCreateDoubleScalar(h->nm));
    mxSetField(plhs[0], 0, "nsph", mxCreateDoubleScalar(h->nsph));
    mxSetField(plhs[0], 0, "batch", mxCreateDoubleScalar(h->B));
    mxArray* idx = new_i32(1, (size_t)h->njoints);
    if (rmx_model_idxR(rmx_group_shard_model(h->g, 0), (int*)mxGetData(idx))) die_rmx("rmx_model_idxR");
    mxSetField(plhs[0], 0, "idxR", idx);
    mxSetField(plhs[0], 0, "nshards", mxCreateDoubleScalar(h->nshards));
    mxArray* dv = mxCreateDoubleMatrix(1, (size_t)h->nshards, mxREAL);
    mxArray* sf = mxCreateDoubleMatrix(1, (size_t)h->nshards, mxREAL);
    mxArray* sc = mxCreateDoubleMatrix(1, (size_t)h->nshards, mxREAL);
    for (int i = 0; i < h->nshards; ++i) {
        int dev = 0, first = 0, count = 0;
        if (rmx_group_shard(h->g, i, &dev, &first, &count)) die_rmx("rmx_group_shard");
        mxGetPr(dv)[i] = dev; mxGetPr(sf)[i] = first; mxGetPr(sc)[i] = count;
    }
    mxSetField(plhs[0], 0, "devices", dv);
    mxSetField(plhs[0], 0, "shard_first", sf);
    mxSetField(plhs[0], 0, "shard_count", sc);
}

static void read_opts(const mxArray* s, rmx_opts* o) {
    if (!s || mxIsEmpty(s)) return;
    if (!mxIsStruct(s)) die("opts must be a struct");
    o->tol = scalar_field(s, "tol", o->tol);
    o->dxMax = scalar_field(s, "dxMax", o->dxMax);
    o->iterMaxPerDof = (int)scalar_field(s, "iterMaxPerDof", o->iterMaxPerDof);
    o->iterLsMax = (int)scalar_field(s, "iterLsMax", o->iterLsMax);
    o->ls_fail_limit = (int)scalar_field(s, "ls_fail_limit", o->ls_fail_limit);
    o->lu_mode = (int)scalar_field(s, "lu_mode", o->lu_mode);
    o->compensated = (int)scalar_field(s, "compensated", o->compensated);
}

/* the outputs of 'step' / 'sync' for nsteps steps: T, V (B x K), st (B x 3 int32), and, when asked for, Q, Qd (nr x B x K) and
 * C (nsph x B x K int32); fills the rmx_stats / rmx_history that point into them */
typedef struct { mxArray *T, *V, *st, *Q, *Qd, *C; rmx_stats stats; rmx_history hist; } step_out_t;
static void step_outputs(const handle_t* h, int nsteps, int want_energy, int want_state, int want_charts, step_out_t* o) {
    const size_t B = (size_t)h->B, K = (size_t)nsteps;
    memset(o, 0, sizeof *o);
    o->T = mxCreateDoubleMatrix(B, want_energy ? K : 0, mxREAL);
    o->V = mxCreateDoubleMatrix(B, want_energy ? K : 0, mxREAL);
    o->st = new_i32(B, 3);
    int* sp = (int*)mxGetData(o->st);
    o->stats.newton_iters = sp;
    o->stats.ls_halvings = sp + B;
    o->stats.status = sp + 2 * B;
    o->hist.T = (K && want_energy) ? mxGetPr(o->T) : NULL;
    o->hist.V = (K && want_energy) ? mxGetPr(o->V) : NULL;
    if (want_state) {   /* the full Scene.saveHistory record */
        const mwSize dims[3] = {(mwSize)h->nr, (mwSize)B, (mwSize)K};
        o->Q = mxCreateNumericArray(3, dims, mxDOUBLE_CLASS, mxREAL);
        o->Qd = mxCreateNumericArray(3, dims, mxDOUBLE_CLASS, mxREAL);
        if (K && h->nr) { o->hist.q = mxGetPr(o->Q); o->hist.qdot = mxGetPr(o->Qd); }
    }
    if (want_charts) {   /* JointSpherical.chart after every step: nsph x B x nsteps int32 */
        const mwSize dims[3] = {(mwSize)h->nsph, (mwSize)B, (mwSize)K};
        o->C = mxCreateNumericArray(3, dims, mxINT32_CLASS, mxREAL);
        if (K && h->nsph) o->hist.charts = (int*)mxGetData(o->C);
    }
}
static void step_return(int nlhs, mxArray* plhs[], const step_out_t* o) {
    plhs[0] = o->T;
    if (nlhs > 1) plhs[1] = o->V;
    if (nlhs > 2) plhs[2] = o->st;
    if (nlhs > 3) plhs[3] = o->Q;
    if (nlhs > 4) plhs[4] = o->Qd;
    if (nlhs > 5) plhs[5] = o->C;
}
static int step_args(int nrhs, const mxArray* prhs[], const char* usage, int* itype, rmx_opts* o) {
    if (nrhs < 5) die(usage);
    *itype = (int)mxGetScalar(prhs[2]);
    const int nsteps = (int)mxGetScalar(prhs[4]);
    if (*itype != 1 && *itype != 2) die("itype must be 1 (BDF1) or 2 (BDF2)");
    if (nsteps < 0) die("nsteps < 0");
    rmx_opts_default(o);
    o->h = mxGetScalar(prhs[3]);
    if (nrhs > 5) read_opts(prhs[5], o);
    return nsteps;
}

static void cmd_step(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    handle_t* h = get_handle(nrhs, prhs);
    int itype;
    rmx_opts o;
    const int nsteps = step_args(nrhs, prhs, "usage: [T,V,stats,Q,Qdot,C] = redmax_hip_mex('step', h, itype, hstep, nsteps [, opts])", &itype, &o);
    if (h->pending) die("a 'step_async' of this handle is in flight: 'sync' first");
    step_out_t out;
    step_outputs(h, nsteps, 1, nlhs > 3, nlhs > 5, &out);
    /* simLoop of the whole batch: every shard's launch goes out before the first is waited for (rmx_group_step) */
    if (rmx_group_step(h->g, &o, nsteps, itype, &out.stats, &out.hist)) die_rmx("rmx_group_step");
    step_return(nlhs, plhs, &out);
}

/* redmax_hip_mex('step_async', h, itype, hstep, nsteps [, opts [, record]]): launch and return */
static void cmd_step_async(int nrhs, const mxArray* prhs[]) {
    handle_t* h = get_handle(nrhs, prhs);
    int itype;
    rmx_opts o;
    const int nsteps = step_args(nrhs, prhs, "usage: redmax_hip_mex('step_async', h, itype, hstep, nsteps [, opts [, record]])", &itype, &o);
    if (h->pending) die("a 'step_async' of this handle is already in flight: 'sync' first");
    const int record = nrhs > 6 ? (int)mxGetScalar(prhs[6]) : RMX_REC_ENERGY;
    if (rmx_group_step_async(h->g, &o, nsteps, itype, record)) {
        char msg[512];
        strncpy(msg, rmx_last_error(), sizeof msg - 1);
        msg[sizeof msg - 1] = 0;
        rmx_group_sync(h->g, NULL, NULL);      /* drain the shards that did start */
        mexErrMsgIdAndTxt("redmax:hip", "rmx_group_step_async: %s", msg);
    }
    h->pending = 1;
    h->pending_nsteps = nsteps;
    h->pending_record = record;
}

/* [T,V,st,Q,Qd,C] = redmax_hip_mex('sync', h): wait for the launches of 'step_async' and gather */
static void cmd_sync(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    handle_t* h = get_handle(nrhs, prhs);
    if (!h->pending) die("'sync' without a 'step_async' in flight");
    const int rec = h->pending_record;
    if (nlhs > 3 && !(rec & RMX_REC_STATE)) die("'sync': Q, Qdot were not recorded (record bit 2 of 'step_async')");
    if (nlhs > 5 && !(rec & RMX_REC_CHARTS)) die("'sync': the charts were not recorded (record bit 4 of 'step_async')");
    step_out_t out;
    step_outputs(h, h->pending_nsteps, rec & RMX_REC_ENERGY, nlhs > 3, nlhs > 5, &out);
    h->pending = 0;
    if (rmx_group_sync(h->g, &out.stats, &out.hist)) die_rmx("rmx_group_sync");
    step_return(nlhs, plhs, &out);
}

static void cmd_euler(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    handle_t* h = get_handle(nrhs, prhs);
    if (nrhs < 4) die("usage: [T,V] = redmax_hip_mex('euler', h, hstep, nsteps)");
    const int nsteps = (int)mxGetScalar(prhs[3]);
    if (nsteps < 0) die("nsteps < 0");
    mxArray* T = mxCreateDoubleMatrix((size_t)h->B, (size_t)nsteps, mxREAL);
    mxArray* V = mxCreateDoubleMatrix((size_t)h->B, (size_t)nsteps, mxREAL);
    if (h->nshards == 1) {
        size_t f;
        if (rmx_step_euler(shard(h, 0, &f), mxGetScalar(prhs[2]), nsteps, nsteps ? mxGetPr(T) : NULL, nsteps ? mxGetPr(V) : NULL)) die_rmx("rmx_step_euler");
    } else {      /* rmx_step_euler writes dense [nsteps][shard] arrays: per shard into a scratch matrix, then into the columns */
        for (int s = 0; s < h->nshards; ++s) {
            size_t f;
            int cnt = 0;
            rmx_batch* b = shard(h, s, &f);
            rmx_group_shard(h->g, s, NULL, NULL, &cnt);
            double* tmp = (double*)mxCalloc(2 * (size_t)cnt * (size_t)nsteps + 1, sizeof(double));
            double* tv = tmp + (size_t)cnt * (size_t)nsteps;
            if (rmx_step_euler(b, mxGetScalar(prhs[2]), nsteps, nsteps ? tmp : NULL, nsteps ? tv : NULL)) { mxFree(tmp); die_rmx("rmx_step_euler"); }
            for (int k = 0; k < nsteps; ++k) {
                memcpy(mxGetPr(T) + (size_t)k * (size_t)h->B + f, tmp + (size_t)k * (size_t)cnt, sizeof(double) * (size_t)cnt);
                memcpy(mxGetPr(V) + (size_t)k * (size_t)h->B + f, tv + (size_t)k * (size_t)cnt, sizeof(double) * (size_t)cnt);
            }
            mxFree(tmp);
        }
    }
    plhs[0] = T;
    if (nlhs > 1) plhs[1] = V;
}

static void cmd_eval(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    handle_t* h = get_handle(nrhs, prhs);
    if (nrhs < 6) die("usage: [g,H] = redmax_hip_mex('eval', h, q, qA, qB, eta)");
    const double* q = state_arg(prhs[2], h, "q");
    const double* qA = state_arg(prhs[3], h, "qA");
    const double* qB = state_arg(prhs[4], h, "qB");
    mxArray* g = mxCreateDoubleMatrix((size_t)h->nr, (size_t)h->B, mxREAL);
    mxArray* H = NULL;
    if (nlhs > 1) {   /* nargout == 1 selects the residual-only path, as in evalBDF1 (driverRedMaxBDF1.m:165) */
        const mwSize dims[3] = {(mwSize)h->nr, (mwSize)h->nr, (mwSize)h->B};
        H = mxCreateNumericArray(3, dims, mxDOUBLE_CLASS, mxREAL);
    }
    for (int s = 0; s < h->nshards; ++s) {
        size_t f;
        rmx_batch* b = shard(h, s, &f);
        const size_t o = f * (size_t)h->nr;
        if (rmx_eval(b, q + o, qA + o, qB + o, mxGetScalar(prhs[5]), mxGetPr(g) + o, H ? mxGetPr(H) + o * (size_t)h->nr : NULL)) die_rmx("rmx_eval");
    }
    plhs[0] = g;
    if (nlhs > 1) plhs[1] = H;
}

/* [M,f,K,D,dMv] = redmax_hip_mex('values', h, q, qdot [, v]) - computeValues' full output (driverRedMaxBDF1.m:188-243: [M,f,dMdq,K,D])
 * at (q, qdot): M, K, D nr x nr x B, f nr x B; with v (nr x B) also dMv, whose column i is dMdq(:,:,i) v (evalBDF1 :181-184 uses the
 * tensor in exactly this form, v = dqtmp).  rmx_compute_values, per shard. */
static void cmd_values(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    handle_t* h = get_handle(nrhs, prhs);
    if (nrhs < 4) die("usage: [M,f,K,D,dMv] = redmax_hip_mex('values', h, q, qdot [, v])");
    const double* q = state_arg(prhs[2], h, "q");
    const double* qd = state_arg(prhs[3], h, "qdot");
    const double* v = nrhs > 4 ? state_arg(prhs[4], h, "v") : NULL;
    if (nlhs > 4 && !v) die("values: the fifth output dMv needs v");
    const mwSize dims[3] = {(mwSize)h->nr, (mwSize)h->nr, (mwSize)h->B};
    mxArray* out[5] = {NULL, NULL, NULL, NULL, NULL};
    out[0] = mxCreateNumericArray(3, dims, mxDOUBLE_CLASS, mxREAL);
    out[1] = mxCreateDoubleMatrix((size_t)h->nr, (size_t)h->B, mxREAL);
    if (nlhs > 2) out[2] = mxCreateNumericArray(3, dims, mxDOUBLE_CLASS, mxREAL);
    if (nlhs > 3) out[3] = mxCreateNumericArray(3, dims, mxDOUBLE_CLASS, mxREAL);
    if (nlhs > 4) out[4] = mxCreateNumericArray(3, dims, mxDOUBLE_CLASS, mxREAL);
    for (int s = 0; s < h->nshards; ++s) {
        size_t f;
        rmx_batch* b = shard(h, s, &f);
        const size_t o = f * (size_t)h->nr, oo = o * (size_t)h->nr;
        if (rmx_compute_values(b, q + o, qd + o, v ? v + o : NULL, mxGetPr(out[0]) + oo, mxGetPr(out[1]) + o,
                               out[3] ? mxGetPr(out[3]) + oo : NULL, out[2] ? mxGetPr(out[2]) + oo : NULL,
                               out[4] ? mxGetPr(out[4]) + oo : NULL, NULL))
            die_rmx("rmx_compute_values");
    }
    for (int i = 0; i < 5 && (i == 0 || i < nlhs); ++i) plhs[i] = out[i];
}

static void cmd_adjoint(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    handle_t* h = get_handle(nrhs, prhs);
    if (nrhs < 6) die("usage: [P,dPdp,stats] = redmax_hip_mex('adjoint', h, hstep, nsteps, task, p [, integrator])");
    const mxArray* t = prhs[4];
    if (!mxIsStruct(t)) die("task must be a struct");
    rmx_task_pointpos task;
    memset(&task, 0, sizeof task);
    task.body = (int)scalar_field(t, "body", 1) - 1;   /* MATLAB listing index -> 0-based */
    const mxArray* xl = field(t, "xlocal", 1);
    const mxArray* xt = field(t, "xtarget", 1);
    if (mxGetNumberOfElements(xl) != 3 || mxGetNumberOfElements(xt) != 3) die("task.xlocal / task.xtarget must have 3 elements");
    memcpy(task.xlocal, mxGetPr(xl), 3 * sizeof(double));
    memcpy(task.xtarget, mxGetPr(xt), 3 * sizeof(double));
    task.step = (int)scalar_field(t, "step", 0);
    task.pscale = scalar_field(t, "pscale", 1.0);
    task.wreg = scalar_field(t, "wreg", 0.0);
    task.wpos = scalar_field(t, "wpos", 1.0);
    rmx_opts o;
    rmx_opts_default(&o);
    o.h = mxGetScalar(prhs[2]);
    o.iterMaxPerDof = 5;                               /* driverRedMaxAdjointBDF1.m:108 */
    const int nsteps = (int)mxGetScalar(prhs[3]);
    const double* p = state_arg(prhs[5], h, "p");
    mxArray* P = mxCreateDoubleMatrix(1, (size_t)h->B, mxREAL);
    mxArray* dPdp = mxCreateDoubleMatrix((size_t)h->nr, (size_t)h->B, mxREAL);
    mxArray* st = new_i32((size_t)h->B, 2);
    int* sp = (int*)mxGetData(st);
    rmx_stats stats;
    stats.newton_iters = sp;
    stats.ls_halvings = NULL;
    stats.status = sp + h->B;
    /* optional 7th argument: the integrator, 1 = BDF1 (driverRedMaxAdjointBDF1.m, default), 2 = BDF2 (driverRedMaxAdjointBDF2.m) */
    const int integ = nrhs > 6 ? (int)mxGetScalar(prhs[6]) : 1;
    if (integ != 1 && integ != 2) die("adjoint: the integrator must be 1 (BDF1) or 2 (BDF2)");
    for (int s = 0; s < h->nshards; ++s) {     /* shard by shard (the adjoint has no asynchronous entry: its outputs are per-call host buffers) */
        size_t f;
        rmx_batch* b = shard(h, s, &f);
        rmx_stats st_s;
        st_s.newton_iters = stats.newton_iters + f;
        st_s.ls_halvings = NULL;
        st_s.status = stats.status + f;
        if ((integ == 1 ? rmx_adjoint_bdf1 : rmx_adjoint_bdf2)(b, &o, nsteps, &task, p + f * (size_t)h->nr, mxGetPr(P) + f,
                                                                mxGetPr(dPdp) + f * (size_t)h->nr, &st_s))
            die_rmx(integ == 1 ? "rmx_adjoint_bdf1" : "rmx_adjoint_bdf2");
    }
    plhs[0] = P;
    if (nlhs > 1) plhs[1] = dPdp;
    if (nlhs > 2) plhs[2] = st;
}

/* `clear mex` / MATLAB exit: free what is still alive on the device */
static void at_exit(void) {
    for (int i = 0; i < MAX_LIVE; ++i)
        if (g_live[i]) {
            rmx_group_destroy(g_live[i]->g);
            mxFree(g_live[i]);
            g_live[i] = NULL;
        }
}

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    static int registered = 0;
    if (!registered) {
        mexAtExit(at_exit);
        registered = 1;
    }
    char cmd[24];      /* the command table: tests/test_mex_gateway.py checks that matlab/+redmax/HipSim.m uses these and only these */
    if (nrhs < 1 || mxGetString(prhs[0], cmd, sizeof cmd)) die("first argument must be a command string");
    /* include/redmax_hip.h "Asynchronous stepping": between a 'step_async' and its 'sync' only 'sync', 'timing', 'info' and 'destroy'
     * may name the handle ('step' / 'step_async' refuse with their own text).  Everything below reads or writes the state, the
     * scratch buffers or the counters of a launch in flight, and would clear its pending mark without taking the event time. */
    static const char* const needs_idle[] = {"set", "get", "gather", "euler", "eval", "values", "energy", "getcharts", "setcharts", "ticks",
                                             "adjoint", NULL};
    for (int i = 0; needs_idle[i]; ++i)
        if (!strcmp(cmd, needs_idle[i]) && get_handle(nrhs, prhs)->pending)
            mexErrMsgIdAndTxt("redmax:hip", "'%s' while a 'step_async' of this handle is in flight: 'sync' first", cmd);
    if (!strcmp(cmd, "version")) {
        plhs[0] = mxCreateDoubleScalar((double)rmx_version());
    } else if (!strcmp(cmd, "devices")) {
        plhs[0] = mxCreateDoubleScalar((double)rmx_device_count());
    } else if (!strcmp(cmd, "create")) {
        cmd_create(nlhs, plhs, nrhs, prhs);
    } else if (!strcmp(cmd, "destroy")) {
        cmd_destroy(nrhs, prhs);
    } else if (!strcmp(cmd, "info")) {
        cmd_info(plhs, nrhs, prhs);
    } else if (!strcmp(cmd, "set")) {
        handle_t* h = get_handle(nrhs, prhs);
        if (nrhs < 4) die("usage: redmax_hip_mex('set', h, q, qdot)");
        if (rmx_group_set_state(h->g, state_arg(prhs[2], h, "q"), state_arg(prhs[3], h, "qdot"))) die_rmx("rmx_group_set_state");
    } else if (!strcmp(cmd, "get")) {
        handle_t* h = get_handle(nrhs, prhs);
        mxArray* q = mxCreateDoubleMatrix((size_t)h->nr, (size_t)h->B, mxREAL);
        mxArray* qd = mxCreateDoubleMatrix((size_t)h->nr, (size_t)h->B, mxREAL);
        if (rmx_group_get_state(h->g, mxGetPr(q), mxGetPr(qd))) die_rmx("rmx_group_get_state");
        plhs[0] = q;
        if (nlhs > 1) plhs[1] = qd;
    } else if (!strcmp(cmd, "step")) {
        cmd_step(nlhs, plhs, nrhs, prhs);
    } else if (!strcmp(cmd, "step_async")) {
        cmd_step_async(nrhs, prhs);
    } else if (!strcmp(cmd, "sync")) {
        cmd_sync(nlhs, plhs, nrhs, prhs);
    } else if (!strcmp(cmd, "timing")) {
        handle_t* h = get_handle(nrhs, prhs);
        mxArray* k = mxCreateDoubleMatrix(1, (size_t)h->nshards, mxREAL);
        mxArray* t0 = mxCreateDoubleMatrix(1, (size_t)h->nshards, mxREAL);
        mxArray* t1 = mxCreateDoubleMatrix(1, (size_t)h->nshards, mxREAL);
        double wall = 0.0;
        if (rmx_group_timing(h->g, &wall, mxGetPr(k), mxGetPr(t0), mxGetPr(t1))) die_rmx("rmx_group_timing");
        plhs[0] = mxCreateDoubleScalar(wall);
        if (nlhs > 1) plhs[1] = k;
        if (nlhs > 2) plhs[2] = t0;
        if (nlhs > 3) plhs[3] = t1;
    } else if (!strcmp(cmd, "euler")) {
        cmd_euler(nlhs, plhs, nrhs, prhs);
    } else if (!strcmp(cmd, "eval")) {
        cmd_eval(nlhs, plhs, nrhs, prhs);
    } else if (!strcmp(cmd, "values")) {
        cmd_values(nlhs, plhs, nrhs, prhs);
    } else if (!strcmp(cmd, "energy")) {
        handle_t* h = get_handle(nrhs, prhs);
        mxArray* T = mxCreateDoubleMatrix(1, (size_t)h->B, mxREAL);
        mxArray* V = mxCreateDoubleMatrix(1, (size_t)h->B, mxREAL);
        if (rmx_group_energy(h->g, mxGetPr(T), mxGetPr(V))) die_rmx("rmx_group_energy");
        plhs[0] = T;
        if (nlhs > 1) plhs[1] = V;
    } else if (!strcmp(cmd, "gather")) {
        /* [q, qdot, path] = redmax_hip_mex('gather', h [, root]): the final gather of the sharded batch with DEVICE-resident destinations
         * (rmx_group_gather: RCCL over the group's devices, copies when a device is listed twice) - every shard's device (root omitted or
         * < 0) or shard `root`'s device alone ends up with the whole batch; q, qdot: that device copy read back (nr x batch), what a host
         * with gpuArray support would wrap in place (rmx_group_gathered); path: how the gather travelled (rmx_group_gather_path) */
        handle_t* h = get_handle(nrhs, prhs);
        const int root = (nrhs > 2 && mxGetScalar(prhs[2]) >= 0.0) ? (int)mxGetScalar(prhs[2]) : RMX_GATHER_ALL;
        if (rmx_group_gather(h->g, root)) die_rmx("rmx_group_gather");
        mxArray* q = mxCreateDoubleMatrix((size_t)h->nr, (size_t)h->B, mxREAL);
        mxArray* qd = mxCreateDoubleMatrix((size_t)h->nr, (size_t)h->B, mxREAL);
        if (rmx_group_gathered_read(h->g, root >= 0 ? root : 0, mxGetPr(q), mxGetPr(qd))) die_rmx("rmx_group_gathered_read");
        plhs[0] = q;
        if (nlhs > 1) plhs[1] = qd;
        if (nlhs > 2) plhs[2] = mxCreateString(rmx_group_gather_path(h->g));
    } else if (!strcmp(cmd, "getcharts")) {
        handle_t* h = get_handle(nrhs, prhs);
        mxArray* c = new_i32((size_t)h->nsph, (size_t)h->B);
        for (int s = 0; s < h->nshards && h->nsph; ++s) {
            size_t f;
            rmx_batch* b = shard(h, s, &f);
            if (rmx_get_charts(b, (int*)mxGetData(c) + f * (size_t)h->nsph)) die_rmx("rmx_get_charts");
        }
        plhs[0] = c;
    } else if (!strcmp(cmd, "setcharts")) {
        handle_t* h = get_handle(nrhs, prhs);
        if (nrhs < 3 || !mxIsInt32(prhs[2]) || mxGetNumberOfElements(prhs[2]) != (size_t)h->nsph * (size_t)h->B)
            die("charts must be an int32 nsph x batch array");
        for (int s = 0; s < h->nshards && h->nsph; ++s) {
            size_t f;
            rmx_batch* b = shard(h, s, &f);
            if (rmx_set_charts(b, (const int*)mxGetData(prhs[2]) + f * (size_t)h->nsph)) die_rmx("rmx_set_charts");
        }
    } else if (!strcmp(cmd, "ticks")) {      /* t = redmax_hip_mex('ticks', h): per-rollout share of the last step launch (rmx_step_ticks) */
        handle_t* h = get_handle(nrhs, prhs);
        unsigned long long* t = (unsigned long long*)mxCalloc((size_t)h->B, sizeof *t);
        for (int s = 0; s < h->nshards; ++s) {
            size_t f;
            rmx_batch* b = shard(h, s, &f);
            if (rmx_step_ticks(b, t + f)) { mxFree(t); die_rmx("rmx_step_ticks"); }
        }
        mxArray* out = mxCreateDoubleMatrix(1, (size_t)h->B, mxREAL);
        for (int i = 0; i < h->B; ++i) mxGetPr(out)[i] = (double)t[i];
        mxFree(t);
        plhs[0] = out;
    } else if (!strcmp(cmd, "adjoint")) {
        cmd_adjoint(nlhs, plhs, nrhs, prhs);
    } else {
        mexErrMsgIdAndTxt("redmax:hip", "unknown command '%s'", cmd);
    }
}
