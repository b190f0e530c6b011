// redmax_hip.hip -- the C ABI of include/redmax_hip.h (host side).
//
// Scene.init()-equivalent flattening of the scene listing into SoA device constants (matlab-diff/+redmax/Scene.m:59-119),
// batch state in HBM, streams/events and launch plumbing.  Kernels and their launchers: rmx_kernels.hip (one object per
// padded tree size, declared in rmx_host.h); device code: rmx_device.h.
// There is no CPU fallback: every entry point needs a HIP device.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <cstdlib>
#include <string>
#include <vector>

#include "rmx_host.h"

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



extern "C" const char* rmx_last_error(void) { return g_err.c_str(); }
static void hist_free(rmx_batch* b);
static hipError_t set_started(rmx_batch* b, int v);

// An error already pending in this thread's HIP state when an entry point is about to launch.  If this batch still has an
// asynchronous launch of its own that nobody waited for (rmx_step_bdf1_async without rmx_sync), the error is reported as that
// launch's and nothing is launched on top of it.  Otherwise it was left behind by another user of HIP in this process: it is taken
// out of this launch's verdict (hipGetLastError clears it), but not silently - rmx_last_error() keeps the note.
static int pending_error_check(rmx_batch* b, const char* who) {
    const hipError_t prev = hipGetLastError();
    if (prev == hipSuccess) return RMX_OK;
    if (b && b->async_pending)
        return fail(RMX_E_HIP, std::string(who) + ": an earlier asynchronous launch of this batch failed: " + hipGetErrorString(prev));
    g_err = std::string(who) + ": note: a HIP error was already pending before this call (left by another user of the process): " + hipGetErrorString(prev);
    return RMX_OK;
}
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
    o->compensated = 1;
    o->ls_fail_limit = 0;   // the reference: never cut the Newton loop short
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

// Scene.init() for a listing of fixed / revolute / prismatic joints.  idx_explicit (or NULL): reduced index of every listed
// joint, given when the listing is the lowered form of a scene with multi-DOF joints (rmx_model_create below).
static int model_create_flat(const rmx_model_desc* d, const int* idx_explicit, const std::vector<int>* sph_first, int device,
                             rmx_model** out) {
    if (!d || !out) return fail(RMX_E_INVALID, "null argument");
    *out = nullptr;
    int ndev = rmx_device_count();
    if (ndev <= 0) return fail(RMX_E_NODEVICE, "no HIP device visible: redmax_hip has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(RMX_E_INVALID, "device index out of range");
    const int n = d->njoints;
    if (n < 1 || n > BIG_MAXN)
        return fail(RMX_E_INVALID, "the scene needs " + std::to_string(n) + " 1-DOF nodes; the limit is " + std::to_string(BIG_MAXN) +
                                       " (up to " + std::to_string(MAXN) + ": one wavefront per tree; beyond: one workgroup per tree)");
    if (!d->parent || !d->type || !d->axis || !d->E0_pj || !d->E0_ji || !d->I_i)
        return fail(RMX_E_INVALID, "parent/type/axis/E0_pj/E0_ji/I_i are required");
    // node stride of the SoA constant arrays and rows of the ancestor table: the one-wavefront kernels (n <= 64) and the
    // one-workgroup kernels for larger trees (rmx_big.hip)
    const bool big = n > MAXN;
    const int NS = big ? BIG_MAXN : MAXN;
    const int NROUNDS = big ? BIG_MAXROUNDS : MAXROUNDS;
    // ---- validate the listing: exactly one root first, parents before children (Scene.m:66-67)
    for (int i = 0; i < n; ++i) {
        if (d->type[i] < 0 || d->type[i] > 2) return fail(RMX_E_INVALID, "unsupported joint type");
        if (i == 0 && d->parent[i] != -1) return fail(RMX_E_INVALID, "joint 0 must be the root");
        if (i > 0 && (d->parent[i] < 0 || d->parent[i] >= i)) return fail(RMX_E_INVALID, "joints must be listed parent-before-child with a single root");
    }
    // ---- reduced / maximal numbering, leaf-to-root over the LISTING (Scene.m:69-71, Joint.countDofs, Body.countDofs)
    std::vector<int> idxL(n, -1);
    int nr = 0;
    for (int i = n - 1; i >= 0; --i)
        if (d->type[i] != RMX_JOINT_FIXED) idxL[i] = idx_explicit ? idx_explicit[i] : nr, ++nr;
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
    if (rounds > NROUNDS) return fail(RMX_E_INVALID, "tree too deep");

    // constants of listed joint L for joint type jtype about axis_in: Kc[36] = rows R(9),p(3) of K0,K1,K2 ; sbc[6] = A0_ij S
    auto joint_consts = [&](const int L, const int jtype, const double* axis_in, double* Kc, double* sbc) -> bool {
        const M4 E0_pj = from_cm(d->E0_pj + 16 * L);
        const M4 E0_ji = from_cm(d->E0_ji + 16 * L);
        M4 Lm = E0_pj;   // root: E_wj = E0_pj Q  (Joint.m:404-415)
        if (d->parent[L] >= 0) Lm = mul(inv(from_cm(d->E0_ji + 16 * d->parent[L])), E0_pj);   // parent body -> joint frame
        double ax[3] = {axis_in[0], axis_in[1], axis_in[2]};
        if (jtype != RMX_JOINT_FIXED) {   // JointRevolute.m:14 / JointPrismatic.m:15
            double nn = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
            if (!(nn > 0)) return false;
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
        if (jtype == RMX_JOINT_REVOLUTE) {
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
            if (jtype == RMX_JOINT_PRISMATIC) m3v(LR, ax, K1p);   // p(q) = a q  (JointPrismatic.m:29-33)
        }
        for (int c = 0; c < 9; ++c) {
            Kc[c] = K0R[c];
            Kc[12 + c] = K1R[c];
            Kc[24 + c] = K2R[c];
        }
        for (int c = 0; c < 3; ++c) {
            Kc[9 + c] = K0p[c];
            Kc[21 + c] = K1p[c];
            Kc[33 + c] = K2p[c];
        }
        // body-frame joint screw A0_ij S  (Body.setBodyTransform Body.m:46-51, Joint.m:508): Ad(E0_ij) [w; v]
        {
            const M4 E0_ij = inv(E0_ji);
            double S[6] = {0, 0, 0, 0, 0, 0};
            if (jtype == RMX_JOINT_REVOLUTE) { S[0] = ax[0]; S[1] = ax[1]; S[2] = ax[2]; }
            if (jtype == RMX_JOINT_PRISMATIC) { S[3] = ax[0]; S[4] = ax[1]; S[5] = ax[2]; }
            double Rm[9], pm[3] = {E0_ij.a[0][3], E0_ij.a[1][3], E0_ij.a[2][3]};
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) Rm[3 * i + j] = E0_ij.a[i][j];
            double w3[3], v3[3];
            m3v(Rm, S, w3);
            m3v(Rm, S + 3, v3);
            const double cx[3] = {pm[1] * w3[2] - pm[2] * w3[1], pm[2] * w3[0] - pm[0] * w3[2], pm[0] * w3[1] - pm[1] * w3[0]};
            for (int c = 0; c < 3; ++c) {
                sbc[c] = w3[c];
                sbc[3 + c] = v3[c] + cx[c];
            }
        }
        return true;
    };
    // ---- constants per node
    std::vector<double> K(36 * NS, 0.0), sb(6 * NS, 0.0), I4(4 * NS, 0.0), prm(8 * NS, 0.0);
    std::vector<int> type(NS, 0), idx(NS, -1), endd(NS, 0), anc(NROUNDS * NS, -1);
    std::vector<unsigned long long> rel(2 * MAXN, 0ull);      // relation bit masks: trees of up to 64 nodes only
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
                if (!big) {
                    rel[k] |= 1ull << t;            // t is a strict ancestor of k
                    rel[MAXN + t] |= 1ull << k;     // k is a strict descendant of t
                }
            }
            for (int r = 0; r < NROUNDS; ++r) {
                int lv = 1 << r;
                anc[r * NS + k] = (lv <= (int)chain.size()) ? chain[lv - 1] : -1;
            }
        }
        (void)a;
        {
            double Kc[36], sbc[6];
            if (!joint_consts(L, type[k], d->axis + 3 * L, Kc, sbc)) return fail(RMX_E_INVALID, "zero joint axis");
            for (int c = 0; c < 36; ++c) K[c * NS + k] = Kc[c];
            for (int c = 0; c < 6; ++c) sb[c * NS + k] = sbc[c];
        }
        // inertia: the reference allows a general diagonal, but mass entries must agree (Body.m:107 uses M_i(4,4))
        const double* Ii = d->I_i + 6 * L;
        I4[0 * NS + k] = Ii[0];
        I4[1 * NS + k] = Ii[1];
        I4[2 * NS + k] = Ii[2];
        I4[3 * NS + k] = Ii[3];
        if (Ii[3] != Ii[4] || Ii[3] != Ii[5]) return fail(RMX_E_INVALID, "I_i(4:6) must all equal the body mass");
        prm[0 * NS + k] = d->tau ? d->tau[L] : 0.0;
        prm[1 * NS + k] = d->stiffness ? d->stiffness[L] : 0.0;
        prm[2 * NS + k] = d->damping ? d->damping[L] : 0.0;
        prm[3 * NS + k] = d->qRest ? d->qRest[L] : 0.0;
        prm[4 * NS + k] = d->qLimL ? d->qLimL[L] : -1e8;   // Joint.m:77-80 defaults
        prm[5 * NS + k] = d->qLimU ? d->qLimU[L] : 1e8;
        prm[6 * NS + k] = d->qLimK ? d->qLimK[L] : 1e8;
        prm[7 * NS + k] = d->qLimD ? d->qLimD[L] : 0.0;
    }

    // axis variants of the spherical group nodes: chart switches (JointSpherical.reparam_) swap them into LDS on the device
    const int nsph = sph_first ? (int)sph_first->size() : 0;
    if (nsph > MAXSPH) return fail(RMX_E_INVALID, "too many spherical joints: a scene may hold at most " + std::to_string(MAXSPH) + " JointSpherical / JointFree3D joints (their Euler-chart tables are sized for that)");
    std::vector<double> sphV((size_t)nsph * 9 * SPH_ROWS + 1, 0.0);
    for (int g = 0; g < nsph; ++g)
        for (int k = 0; k < 3; ++k)
            for (int a = 0; a < 3; ++a) {
                const double ea[3] = {a == 0 ? 1.0 : 0.0, a == 1 ? 1.0 : 0.0, a == 2 ? 1.0 : 0.0};
                double* v = &sphV[((size_t)(g * 3 + k) * 3 + a) * SPH_ROWS];
                if (pos[(*sph_first)[g] + k] != pos[(*sph_first)[g]] + k) return fail(RMX_E_INVALID, "spherical group is not contiguous");
                joint_consts((*sph_first)[g] + k, RMX_JOINT_REVOLUTE, ea, v, v + 36);
            }

    rmx_model* m = new rmx_model();
    m->device = device;
    m->n = n;
    m->nr = nr;
    m->nlist = n;
    m->nm = 6 * n;
    m->idx_listing = idxL;
    m->node_of_listing = pos;
    m->NP = n <= 4 ? 4 : n <= 8 ? 8 : n <= 16 ? 16 : n <= 32 ? 32 : n <= 64 ? 64 : BIG_MAXN;
    m->big = big;
    // per-node constants + the rows of the per-body ground frames (con_setup; present in every launch so that attaching a
    // ForceGroundCuboid later does not change the layout)
    m->smem_bytes = big ? 0 : sizeof(double) * ((size_t)acc_doubles(n, m->NP) + (size_t)(NCONST + NGROUND) * cstride(m->NP));
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
            m->n_simd = 4 * prop.multiProcessorCount;
            m->lds_limit = (int)prop.sharedMemPerBlock;
            // the cooperative groups' timeout, ~2 s in s_memtime ticks: on gfx950 the counter runs at the shader clock (8.9 M ticks per
            // 3.75 ms launch, tools/pairc_ticks.py), on gfx90a / gfx942 at the constant 100 MHz reference clock
            const bool shader_rate = strncmp(prop.gcnArchName, "gfx950", 6) == 0 && prop.clockRate > 0;
            m->coop_ticks = shader_rate ? 2000ull * (unsigned long long)prop.clockRate : 200000000ull;
        }
        if (const char* lim = getenv("RMX_BIG_LDS_LIMIT")) m->lds_limit = atoi(lim);     // development aid (0: H of large trees stays in HBM)
    }
    if (hipSetDevice(device) != hipSuccess) { delete m; return fail(RMX_E_HIP, "hipSetDevice failed"); }
    // the tree as the multifrontal solve of 33..64-node branching trees walks it (DevModel::tree, tree_solve64 in rmx_device.h)
    std::vector<int> tree((size_t)TREE_ROWS * NS, -1);
    int tree_dmax = 0, tree_cmax = 0;
    if (!big && n > 32 && nsph == 0) {
        std::vector<int> dep(n, 0), nch(n, 0);
        bool ok = true;
        for (int k = 0; k < n && ok; ++k) {
            dep[k] = par[k] >= 0 ? dep[par[k]] + 1 : 0;
            if (par[k] >= k) ok = false;                       // (parents first: the depth-first order the kernels rely on)
            if (dep[k] > TREE_DMAX) ok = false;
            tree_dmax = std::max(tree_dmax, dep[k]);
            if (par[k] >= 0) {
                if (nch[par[k]] >= TREE_CMAX) ok = false;
                else tree[(size_t)(1 + TREE_DMAX + nch[par[k]]++) * NS + par[k]] = k;
                tree_cmax = std::max(tree_cmax, nch[par[k]]);
            }
        }
        int roots = 0;
        for (int k = 0; k < n; ++k) roots += par[k] < 0;
        if (roots != 1) ok = false;                            // (one tree: every frontal matrix ends at the same root)
        if (ok) {
            for (int k = 0; k < n; ++k) {
                tree[k] = dep[k];
                { int lv = 1; for (int t = par[k]; t >= 0; t = par[t], ++lv) tree[(size_t)lv * NS + k] = t; }      // row lv: the ancestor lv levels up
                tree[(size_t)(1 + TREE_DMAX + TREE_CMAX) * NS + k] = par[k];
            }
        }
        if (!ok || tree_dmax < 1) tree_dmax = 0;
        const char* ts = getenv("RMX_TREE_SOLVE");             // 0: the dense guarded solve for every tree (tests)
        if (ts && atoi(ts) == 0) tree_dmax = 0;
    }
    const size_t nd = K.size() + sb.size() + I4.size() + prm.size();
    const size_t ni = type.size() + idx.size() + endd.size() + anc.size() + tree.size();
    const size_t bytes = nd * sizeof(double) + rel.size() * sizeof(unsigned long long) + ni * sizeof(int);
    hipError_t e = hipMalloc(&m->dbuf, bytes);
    if (e != hipSuccess) { delete m; return fail(RMX_E_NOMEM, std::string("hipMalloc(model): ") + hipGetErrorString(e)); }
    std::vector<char> host(bytes);
    char* hp = host.data();
    char* dp = (char*)m->dbuf;
    auto put = [&](const void* src, size_t nb) { std::memcpy(hp, src, nb); const void* dptr = dp; hp += nb; dp += nb; return dptr; };
    m->dm.n = n;
    m->dm.nr = nr;
    m->dm.stride = NS;
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
    m->dm.tree = (const int*)put(tree.data(), tree.size() * sizeof(int));
    m->dm.tree_dmax = tree_dmax;
    m->dm.tree_cmax = tree_cmax;
    for (int c = 0; c < 3; ++c) m->dm.grav[c] = d->grav[c];
    e = hipMemcpy(m->dbuf, host.data(), bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(m->dbuf); delete m; return fail(RMX_E_HIP, std::string("hipMemcpy(model): ") + hipGetErrorString(e)); }
    m->dm.nsph = nsph;
    if (nsph) {
        for (int g = 0; g < nsph; ++g) m->dm.sph_first[g] = (short)pos[(*sph_first)[g]];
        e = hipMalloc(&m->dsph, sphV.size() * sizeof(double));
        if (e == hipSuccess) e = hipMemcpy(m->dsph, sphV.data(), sphV.size() * sizeof(double), hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipFree(m->dbuf); if (m->dsph) (void)hipFree(m->dsph); delete m; return fail(RMX_E_HIP, "spherical variant table"); }
        m->dm.sphV = (const double*)m->dsph;
    }
    if (m->NP == 64 && nsph == 0) {
        // Trees of 33..64 nodes: a second copy of the per-node constants, staged as the kernels read them, in global memory.  Batches
        // of more than two rollouts per CU run the kernels that read it from there (33.8 KB of LDS per wavefront instead of 68.6 KB:
        // four wavefronts per CU instead of two); RMX_GCONST_MIN in the environment moves the threshold (tests, measurements).
        e = hipMalloc(&m->dgconst, sizeof(double) * (size_t)NCONST * cstride(64));
        if (e == hipSuccess) {
            launch_stage_consts_64(m, (double*)m->dgconst, nullptr);
            e = hipGetLastError();
            if (e == hipSuccess) e = hipDeviceSynchronize();
        }
        if (e != hipSuccess) {
            std::string msg = std::string("staging the constants table: ") + hipGetErrorString(e);
            rmx_model_destroy(m);
            return fail(RMX_E_HIP, msg);
        }
        m->dm.gconst = (const double*)m->dgconst;
        const char* thr = getenv("RMX_GCONST_MIN");
        m->gconst_min_batch = thr ? atoi(thr) : (m->n_simd > 0 ? m->n_simd / 2 + 1 : 513);
        // ... and batches of at most one rollout per two SIMDs the two-wave kernels (rmx_kernels.hip RMX_PART 5)
        const char* w2 = getenv("RMX_W2_MAX");
        m->w2_max_batch = w2 ? atoi(w2) : (m->n_simd > 0 ? m->n_simd / 2 : 512);
    }
    if (m->NP == 32 && nsph == 0 && m->dm.is_chain && m->dm.n == 32) {
        // the full 32-link chain (BASELINE.json configs[1]) in shards of 128 .. one rollout per two SIMDs: the two-wave kernel of
        // rmx_kernels.hip RMX_PART 6 (RMX_W2_MAX / RMX_W2C_MIN move the bounds)
        const char* w2 = getenv("RMX_W2_MAX");
        const char* w2m = getenv("RMX_W2C_MIN");
        m->w2_max_batch = w2 ? atoi(w2) : (m->n_simd > 0 ? m->n_simd / 2 : 512);
        m->w2_min_batch = w2m ? atoi(w2m) : 128;
    }
    if (m->NP == 16 && !big) m->adj_help_max_batch = m->n_simd > 0 ? m->n_simd / 2 : 512;      // (one rollout per two SIMDs)
    *out = m;
    return RMX_OK;
}

// Scene.init() (Scene.m:59-119).  Multi-DOF joints whose Q(q) is a product of one-parameter motions are lowered to a chain of
// 1-DOF nodes, parent first, the last one carrying the body and the others massless (E0_ji = I, I_i = 0):
//   JointPlanar        (JointPlanar.m:24-31)          prismatic(b1) . prismatic(b2)
//   JointTranslational (JointTranslational.m:22-26)   prismatic(x) . prismatic(y) . prismatic(z)
//   JointUniversal     (JointUniversal.m:71-74)       revolute(x) . revolute(y)            R = X1(q1) Y2(q2)
//   JointFree2D        (JointFree2D.m:20-33)          prismatic(x) . prismatic(y) . revolute(z)   Q = [Rz(q3) [q1;q2;0]]
// Same world transforms, coordinates and velocities, hence the same M, f, K, D; DOF k of joint j keeps the reference's reduced
// index idxR(k) = nr + k with joints counted from the last listed to the first (Joint.countDofs, Joint.m:149-158).
extern "C" int rmx_model_create(const rmx_model_desc* d, int device, rmx_model** out) {
    if (!d || !out) return fail(RMX_E_INVALID, "null argument");
    const int n = d->njoints;
    if (n < 1 || n > BIG_MAXN) return fail(RMX_E_INVALID, "njoints must be in [1," + std::to_string(BIG_MAXN) + "]");
    if (!d->parent || !d->type || !d->axis || !d->E0_pj || !d->E0_ji || !d->I_i)
        return fail(RMX_E_INVALID, "parent/type/axis/E0_pj/E0_ji/I_i are required");
    bool composite = false;
    for (int i = 0; i < n; ++i) {
        if (d->type[i] < 0 || d->type[i] > RMX_JOINT_FREE3D) return fail(RMX_E_INVALID, "unsupported joint type");
        if (d->type[i] > RMX_JOINT_PRISMATIC) composite = true;
        if (i > 0 && (d->parent[i] < 0 || d->parent[i] >= i)) return fail(RMX_E_INVALID, "joints must be listed parent-before-child with a single root");
    }
    if (!composite && !d->qRestR) return model_create_flat(d, nullptr, nullptr, device, out);

    struct Sub { int type; double ax[3]; };
    auto subs_of = [&](int L, std::vector<Sub>& v) -> bool {
        v.clear();
        const double ex[3] = {1, 0, 0}, ey[3] = {0, 1, 0}, ez[3] = {0, 0, 1};
        auto add = [&](int t, const double* a) { v.push_back(Sub{t, {a[0], a[1], a[2]}}); };
        switch (d->type[L]) {
            case RMX_JOINT_PLANAR:
                if (d->plane) { add(RMX_JOINT_PRISMATIC, d->plane + 6 * L); add(RMX_JOINT_PRISMATIC, d->plane + 6 * L + 3); }
                else { add(RMX_JOINT_PRISMATIC, ex); add(RMX_JOINT_PRISMATIC, ey); }     // JointPlanar.m:13-15 default plane
                break;
            case RMX_JOINT_TRANSLATIONAL: add(RMX_JOINT_PRISMATIC, ex); add(RMX_JOINT_PRISMATIC, ey); add(RMX_JOINT_PRISMATIC, ez); break;
            case RMX_JOINT_UNIVERSAL: add(RMX_JOINT_REVOLUTE, ex); add(RMX_JOINT_REVOLUTE, ey); break;
            case RMX_JOINT_FREE2D: add(RMX_JOINT_PRISMATIC, ex); add(RMX_JOINT_PRISMATIC, ey); add(RMX_JOINT_REVOLUTE, ez); break;
            case RMX_JOINT_FREE3D: add(RMX_JOINT_PRISMATIC, ex); add(RMX_JOINT_PRISMATIC, ey); add(RMX_JOINT_PRISMATIC, ez);   // fallthrough
            case RMX_JOINT_SPHERICAL: add(RMX_JOINT_REVOLUTE, ex); add(RMX_JOINT_REVOLUTE, ey); add(RMX_JOINT_REVOLUTE, ez); break;   // chart XYZ
            default: add(d->type[L], d->axis + 3 * L); break;
        }
        return true;
    };
    // reference numbering of every DOF
    std::vector<int> base(n), ndof(n);
    std::vector<Sub> sv;
    int nr = 0;
    for (int L = n - 1; L >= 0; --L) {
        subs_of(L, sv);
        ndof[L] = d->type[L] == RMX_JOINT_FIXED ? 0 : (int)sv.size();
        base[L] = nr;
        nr += ndof[L];
    }
    std::vector<int> parent, type, idx, last(n), sph_first;
    std::vector<double> axis, E0_pj, E0_ji, I_i, qRest, tau, stiff, damp, qLimL, qLimU, qLimK, qLimD;
    const double I16[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    for (int L = 0; L < n; ++L) {
        subs_of(L, sv);
        if (d->type[L] == RMX_JOINT_SPHERICAL || d->type[L] == RMX_JOINT_FREE3D)
            sph_first.push_back((int)type.size() + (d->type[L] == RMX_JOINT_FREE3D ? 3 : 0));
        for (size_t k = 0; k < sv.size(); ++k) {
            const bool fin = k + 1 == sv.size();
            parent.push_back(k == 0 ? (d->parent[L] < 0 ? -1 : last[d->parent[L]]) : (int)type.size() - 1);
            type.push_back(sv[k].type);
            axis.insert(axis.end(), sv[k].ax, sv[k].ax + 3);
            const double* epj = k == 0 ? d->E0_pj + 16 * L : I16;
            const double* eji = fin ? d->E0_ji + 16 * L : I16;
            E0_pj.insert(E0_pj.end(), epj, epj + 16);
            E0_ji.insert(E0_ji.end(), eji, eji + 16);
            for (int c = 0; c < 6; ++c) I_i.push_back(fin ? d->I_i[6 * L + c] : 0.0);
            const int ri = base[L] + (int)k;
            idx.push_back(sv[k].type == RMX_JOINT_FIXED ? -1 : ri);
            qRest.push_back(sv[k].type == RMX_JOINT_FIXED ? 0.0 : d->qRestR ? d->qRestR[ri] : (k == 0 && d->qRest) ? d->qRest[L] : 0.0);
            tau.push_back(d->tau ? d->tau[L] : 0.0);
            stiff.push_back(d->stiffness ? d->stiffness[L] : 0.0);
            damp.push_back(d->damping ? d->damping[L] : 0.0);
            qLimL.push_back(d->qLimL ? d->qLimL[L] : -1e8);
            qLimU.push_back(d->qLimU ? d->qLimU[L] : 1e8);
            qLimK.push_back(d->qLimK ? d->qLimK[L] : 1e8);
            qLimD.push_back(d->qLimD ? d->qLimD[L] : 0.0);
        }
        last[L] = (int)type.size() - 1;
    }
    if ((int)type.size() > BIG_MAXN)
        return fail(RMX_E_INVALID, "the scene needs " + std::to_string(type.size()) + " 1-DOF nodes after lowering its multi-DOF joints; the limit is " + std::to_string(BIG_MAXN));
    rmx_model_desc x{};
    x.njoints = (int)type.size();
    x.parent = parent.data(); x.type = type.data(); x.axis = axis.data();
    x.E0_pj = E0_pj.data(); x.E0_ji = E0_ji.data(); x.I_i = I_i.data();
    x.qRest = qRest.data(); x.tau = tau.data(); x.stiffness = stiff.data(); x.damping = damp.data();
    x.qLimL = qLimL.data(); x.qLimU = qLimU.data(); x.qLimK = qLimK.data(); x.qLimD = qLimD.data();
    for (int c = 0; c < 3; ++c) x.grav[c] = d->grav[c];
    rmx_model* m = nullptr;
    const int rc = model_create_flat(&x, idx.data(), &sph_first, device, &m);
    if (rc) return rc;
    // what the caller sees follows ITS listing: body L is the last node of joint L's chain, idxR(L) its first DOF
    std::vector<int> node_of(n), idxL(n);
    for (int L = 0; L < n; ++L) {
        node_of[L] = m->node_of_listing[last[L]];
        idxL[L] = ndof[L] ? base[L] : -1;
    }
    m->nlist = n;
    m->nm = 6 * n;
    m->node_of_listing = node_of;
    m->idx_listing = idxL;
    *out = m;
    return RMX_OK;
}

extern "C" void rmx_model_destroy(rmx_model* m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    if (m->dbuf) (void)hipFree(m->dbuf);
    if (m->dcon) (void)hipFree(m->dcon);
    if (m->dsph) (void)hipFree(m->dsph);
    if (m->dgconst) (void)hipFree(m->dgconst);
    delete m;
}

// scene.forces{end+1} = ForceGroundCuboid(body); setTransform / setStiffness / setDamping / setFriction
// (scenesRedMax.m:303-309, ForceGroundCuboid.m:18-48) for every flagged body, each with its own ground frame and constants.
extern "C" int rmx_model_set_ground_contact(rmx_model* m, const rmx_ground_contact* gc) {
    if (!m || !gc || !gc->flags || !gc->sides) return fail(RMX_E_INVALID, "null argument");
    if (!(gc->kn >= 0) || !(gc->kt >= 0) || !(gc->mu >= 0) || !(gc->kd >= 0)) return fail(RMX_E_INVALID, "contact constants must be >= 0");
    HIPCHK(hipSetDevice(m->device));
    const int MAXN = m->dm.stride;        // node stride of the table: rmx::MAXN, or BIG_MAXN for the one-workgroup kernels (rmx_big.hip)
    std::vector<double> con((size_t)NCON * MAXN, 0.0);
    bool any = false;
    for (int L = 0; L < m->nlist; ++L) {
        const int k = m->node_of_listing[L];
        con[k] = gc->flags[L] ? 1.0 : 0.0;
        any = any || gc->flags[L];
        for (int c = 0; c < 3; ++c) con[(1 + c) * MAXN + k] = gc->sides[3 * L + c];
        // this body's force object: its ground frame (ng = E(1:3,3), xg = E(1:3,4): ForceGroundCuboid.m:56-57) and constants
        const M4 E = from_cm(gc->E_body ? gc->E_body + 16 * (size_t)L : gc->E);
        const double kn = gc->kn_body ? gc->kn_body[L] : gc->kn, kt = gc->kt_body ? gc->kt_body[L] : gc->kt;
        const double mu = gc->mu_body ? gc->mu_body[L] : gc->mu, kd = gc->kd_body ? gc->kd_body[L] : gc->kd;
        if (gc->flags[L] && (!(kn >= 0) || !(kt >= 0) || !(mu >= 0) || !(kd >= 0))) return fail(RMX_E_INVALID, "contact constants must be >= 0");
        for (int c = 0; c < 3; ++c) {
            con[(4 + c) * MAXN + k] = E.a[c][2];
            con[(7 + c) * MAXN + k] = E.a[c][3];
        }
        con[10 * MAXN + k] = kn;
        con[11 * MAXN + k] = kt;
        con[12 * MAXN + k] = mu;
        con[13 * MAXN + k] = kd;
    }
    // the new table is uploaded first and swapped in only on success; kernels of existing batches that may still read the
    // old one (rmx_step_*_async) are drained before it is freed
    void* fresh = nullptr;
    if (any) {
        HIPCHK(hipMalloc(&fresh, con.size() * sizeof(double)));
        const hipError_t e = hipMemcpy(fresh, con.data(), con.size() * sizeof(double), hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipFree(fresh); return fail(RMX_E_HIP, std::string("hipMemcpy(contact): ") + hipGetErrorString(e)); }
    }
    for (rmx_batch* bb : m->batches) {      // only this model's batches can be reading the old table: other models' streams run on
        const hipError_t es = hipStreamSynchronize(bb->stream);
        if (es != hipSuccess) {
            if (fresh) (void)hipFree(fresh);
            return fail(RMX_E_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(es));
        }
        bb->async_pending = false;
    }
    if (m->dcon) (void)hipFree(m->dcon);
    m->dcon = fresh;
    m->dm.con = (const double*)fresh;
    return RMX_OK;
}
extern "C" int rmx_model_nr(const rmx_model* m) { return m ? m->nr : RMX_E_INVALID; }
extern "C" int rmx_model_nm(const rmx_model* m) { return m ? m->nm : RMX_E_INVALID; }
extern "C" int rmx_model_idxR(const rmx_model* m, int* idx) {
    if (!m || !idx) return fail(RMX_E_INVALID, "null argument");
    for (int i = 0; i < m->nlist; ++i) idx[i] = m->idx_listing[i];
    return RMX_OK;
}

// Batches that hold cooperative-group buffers (launch_step: serial chains of <= 32 nodes with ForceGroundCuboid), process-wide: the groups of
// ONE launch spin on each other through global memory and must all be resident together, so the SIMDs of a device are DIVIDED among the
// live batches that may run such a launch on it at the same time (two shards of a group on one device, asynchronous BatchSims) - each
// sized as if it owned the device, the partly resident groups of two launches could wait for each other's missing members for ever.
static std::mutex g_coop_mu;
static std::vector<rmx_batch*> g_coop_batches;
static int coop_batches_on_device(int device) {
    std::lock_guard<std::mutex> lk(g_coop_mu);
    int n = 0;
    for (const rmx_batch* b : g_coop_batches) n += (b->m->device == device) ? 1 : 0;
    return n;
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
    alloc((void**)&b->resume, sizeof(int) * batch);
    alloc((void**)&b->ticks, sizeof(unsigned long long) * batch);
    if (m->big) {      // trees of more than 64 nodes: the per-rollout workspace of the one-workgroup-per-tree kernels
        b->bigws_stride = big_ws_doubles(m);
        alloc((void**)&b->bigws, sizeof(double) * b->bigws_stride * (size_t)batch);
    }
    if (m->dm.nsph) {   // every JointSpherical starts in CHART_XYZ (JointSpherical.m:33)
        const std::vector<int> c7((size_t)batch * m->dm.nsph, 7);
        if (e == hipSuccess) e = hipMalloc((void**)&b->chart, c7.size() * sizeof(int));
        if (e == hipSuccess) e = hipMemcpy(b->chart, c7.data(), c7.size() * sizeof(int), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&b->ev0);
    if (e == hipSuccess) e = hipEventCreate(&b->ev1);
    if (e != hipSuccess) {
        std::string msg = std::string("rmx_batch_create: ") + hipGetErrorString(e);
        rmx_batch_destroy(b);
        return fail(RMX_E_HIP, msg);
    }
    m->batches.push_back(b);
    *out = b;
    return RMX_OK;
}

extern "C" void rmx_batch_destroy(rmx_batch* b) {
    if (!b) return;
    (void)hipSetDevice(b->m->device);
    b->m->batches.erase(std::remove(b->m->batches.begin(), b->m->batches.end(), b), b->m->batches.end());
    {
        std::lock_guard<std::mutex> lk(g_coop_mu);
        g_coop_batches.erase(std::remove(g_coop_batches.begin(), g_coop_batches.end(), b), g_coop_batches.end());
    }
    if (b->stream) (void)hipStreamSynchronize(b->stream);
    hist_free(b);
    for (void* p : {(void*)b->q, (void*)b->qd, (void*)b->qp, (void*)b->qdp, (void*)b->tmpA, (void*)b->tmpB, (void*)b->tmpC,
                    (void*)b->started, (void*)b->it, (void*)b->ls, (void*)b->status, (void*)b->resume, (void*)b->park, (void*)b->xch, (void*)b->xrec, b->gargs, (void*)b->chart, (void*)b->ticks, (void*)b->bigws, b->adjws})
        if (p) (void)hipFree(p);
    if (b->ev0) (void)hipEventDestroy(b->ev0);
    if (b->ev1) (void)hipEventDestroy(b->ev1);
    if (b->stream) (void)hipStreamDestroy(b->stream);
    delete b;
}
extern "C" int rmx_batch_size(const rmx_batch* b) { return b ? b->B : RMX_E_INVALID; }
extern "C" void* rmx_batch_stream(const rmx_batch* b) { return b ? (void*)b->stream : nullptr; }
extern "C" double rmx_last_step_ms(const rmx_batch* b) { return b ? b->last_ms : -1.0; }
extern "C" const char* rmx_last_step_kernel(const rmx_batch* b) { return (b && b->last_kernel) ? b->last_kernel : ""; }

static int copy_state(rmx_batch* b, const double* q, const double* qd, hipMemcpyKind kind, bool set) {
    if (!b) return fail(RMX_E_INVALID, "null batch");
    HIPCHK(hipSetDevice(b->m->device));
    const size_t nb = sizeof(double) * (size_t)b->B * b->m->nr;
    if (nb == 0) return RMX_OK;
    if (set) {
        if (q) HIPCHK(hipMemcpyAsync(b->q, q, nb, kind, b->stream));
        if (qd) HIPCHK(hipMemcpyAsync(b->qd, qd, nb, kind, b->stream));
        HIPCHK(set_started(b, 0));   // a new state restarts BDF2 with SDIRK2
        if (b->chart && q) {   // the new coordinates are read in CHART_XYZ (rmx_set_charts afterwards says otherwise)
            const std::vector<int> c7((size_t)b->B * b->m->dm.nsph, 7);
            HIPCHK(hipMemcpyAsync(b->chart, c7.data(), c7.size() * sizeof(int), hipMemcpyHostToDevice, b->stream));
            HIPCHK(hipStreamSynchronize(b->stream));
        }
    } else {
        if (q) HIPCHK(hipMemcpyAsync((void*)q, b->q, nb, kind, b->stream));
        if (qd) HIPCHK(hipMemcpyAsync((void*)qd, b->qd, nb, kind, b->stream));
    }
    HIPCHK(hipStreamSynchronize(b->stream));
    b->async_pending = false;      // the batch's stream has been waited for
    return RMX_OK;
}
extern "C" int rmx_model_nsph(const rmx_model* m) { return m ? m->dm.nsph : RMX_E_INVALID; }
// JointSpherical.chart of every spherical joint and trajectory (reference numbering 1..12); host [batch][nsph]
extern "C" int rmx_get_charts(rmx_batch* b, int* charts) {
    if (!b || !charts) return fail(RMX_E_INVALID, "null argument");
    if (!b->chart) return RMX_OK;
    HIPCHK(hipSetDevice(b->m->device));
    HIPCHK(hipMemcpyAsync(charts, b->chart, sizeof(int) * (size_t)b->B * b->m->dm.nsph, hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    return RMX_OK;
}
extern "C" int rmx_set_charts(rmx_batch* b, const int* charts) {
    if (!b || !charts) return fail(RMX_E_INVALID, "null argument");
    if (!b->chart) return RMX_OK;
    const size_t nc = (size_t)b->B * b->m->dm.nsph;
    for (size_t i = 0; i < nc; ++i)
        if (charts[i] < 1 || charts[i] > 12) return fail(RMX_E_INVALID, "charts must be in 1..12 (JointSpherical.CHART_*)");
    HIPCHK(hipSetDevice(b->m->device));
    HIPCHK(hipMemcpyAsync(b->chart, charts, sizeof(int) * nc, hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    return RMX_OK;
}
extern "C" int rmx_set_state(rmx_batch* b, const double* q, const double* qdot) { return copy_state(b, q, qdot, hipMemcpyHostToDevice, true); }
extern "C" int rmx_get_state(rmx_batch* b, double* q, double* qdot) { return copy_state(b, q, qdot, hipMemcpyDeviceToHost, false); }
extern "C" int rmx_set_state_device(rmx_batch* b, const double* q, const double* qdot) { return copy_state(b, q, qdot, hipMemcpyDeviceToDevice, true); }
extern "C" int rmx_get_state_device(rmx_batch* b, double* q, double* qdot) { return copy_state(b, q, qdot, hipMemcpyDeviceToDevice, false); }

#define DISPATCH_NP(NPV, FN, ...)            \
    switch (NPV) {                            \
        case 4: FN##_4(__VA_ARGS__); break;   \
        case 8: FN##_8(__VA_ARGS__); break;   \
        case 16: FN##_16(__VA_ARGS__); break; \
        case 32: FN##_32(__VA_ARGS__); break; \
        default: FN##_64(__VA_ARGS__); break; \
    }


extern "C" int rmx_eval(rmx_batch* b, const double* q, const double* qA, const double* qB, double eta, double* g, double* H) {
    if (!b || !q || !qA || !qB || !g) return fail(RMX_E_INVALID, "null argument");
    if (!(eta > 0)) return fail(RMX_E_INVALID, "eta must be positive");
    rmx_model* m = b->m;
    HIPCHK(hipSetDevice(m->device));
    if (int rc = pending_error_check(b, "rmx_eval")) return rc;
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
    if (m->big) launch_big_eval(m, b, H != nullptr, eta, dg, dH);
    else DISPATCH_NP(m->NP, launch_eval, m, b, H != nullptr, eta, dg, dH);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(g, dg, nv * sizeof(double), hipMemcpyDeviceToHost, b->stream);
    if (e == hipSuccess && H) e = hipMemcpyAsync(H, dH, nv * m->nr * sizeof(double), hipMemcpyDeviceToHost, b->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
    (void)hipFree(dg);
    if (dH) (void)hipFree(dH);
    if (e != hipSuccess) return fail(RMX_E_HIP, std::string("rmx_eval: ") + hipGetErrorString(e));
    return RMX_OK;
}

extern "C" int rmx_eval_mfd(rmx_batch* b, const double* q, const double* qdot, double* M, double* f, double* D) {
    if (!b || !q || !qdot || !M || !f || !D) return fail(RMX_E_INVALID, "null argument");
    rmx_model* m = b->m;
    if (m->big) return rmx_compute_values(b, q, qdot, nullptr, M, f, D, nullptr, nullptr, nullptr);   // no M / D kernel of their own: from H
    HIPCHK(hipSetDevice(m->device));
    if (int rc = pending_error_check(b, "rmx_eval_mfd")) return rc;
    const size_t nv = (size_t)b->B * m->nr, nn = nv * m->nr;
    if (nv == 0) return RMX_OK;
    double* buf = nullptr;
    HIPCHK(hipMalloc((void**)&buf, (2 * nn + nv) * sizeof(double)));
    double *dM = buf, *dD = buf + nn, *df = buf + 2 * nn;
    hipError_t e = hipMemsetAsync(buf, 0, (2 * nn + nv) * sizeof(double), b->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(b->tmpA, q, nv * sizeof(double), hipMemcpyHostToDevice, b->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(b->tmpB, qdot, nv * sizeof(double), hipMemcpyHostToDevice, b->stream);
    if (e == hipSuccess) {
            DISPATCH_NP(m->NP, launch_mfd, m, b, dM, df, dD);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(M, dM, nn * sizeof(double), hipMemcpyDeviceToHost, b->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(D, dD, nn * sizeof(double), hipMemcpyDeviceToHost, b->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(f, df, nv * sizeof(double), hipMemcpyDeviceToHost, b->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
    (void)hipFree(buf);
    if (e != hipSuccess) return fail(RMX_E_HIP, std::string("rmx_eval_mfd: ") + hipGetErrorString(e));
    return RMX_OK;
}

// computeValues' full output from evaluations of H(eta; v) = M + dMdq v - eta D - eta^2 K at fixed (q, qdot): see include/redmax_hip.h
extern "C" int rmx_compute_values(rmx_batch* b, const double* q, const double* qdot, const double* v, double* M, double* f, double* D,
                                  double* K, double* dMv, double* dMdq) {
    if (!b || !q || !qdot) return fail(RMX_E_INVALID, "null argument");
    if (dMv && !v) return fail(RMX_E_INVALID, "rmx_compute_values: dMv needs v");
    rmx_model* m = b->m;
    const size_t nr = (size_t)m->nr, nv = (size_t)b->B * nr, nn = nv * nr;
    if (nv == 0) return RMX_OK;
    std::vector<double> qA(nv), qB(nv), g(nv), H1(nn), Mt, Dt, ft;
    auto eval_at = [&](const double eta, const double* vv, std::vector<double>& Hout) -> int {
        for (size_t i = 0; i < nv; ++i) {
            qA[i] = q[i] - eta * qdot[i];
            qB[i] = vv ? q[i] - vv[i] : q[i];
        }
        return rmx_eval(b, q, qA.data(), qB.data(), eta, g.data(), Hout.data());
    };
    if (int rc = eval_at(1.0, nullptr, H1)) return rc;           // H1 = M - D - K ; g = -f
    if (f) for (size_t i = 0; i < nv; ++i) f[i] = -g[i];
    if (M || D || K) {
        if (!M) { Mt.resize(nn); M = Mt.data(); }
        if (!D) { Dt.resize(nn); D = Dt.data(); }
        if (!m->big) {
            ft.resize(nv);
            if (int rc = rmx_eval_mfd(b, q, qdot, M, ft.data(), D)) return rc;
            if (K) for (size_t i = 0; i < nn; ++i) K[i] = (M[i] - D[i]) - H1[i];
        } else {
            // No M / D kernel at this size: the pieces are taken apart from H(e), H(2 e), H(e / 2):
            //   3 H(e) - H(2 e) - 2 H(e / 2) = 3/2 e^2 K ;  H(e) - H(2 e) = e D + 3 e^2 K ;  M = H(e) + e D + e^2 K.
            // H(e) carries roundoff of order eps (|M| + e |D| + e^2 |K|), so a piece comes out with good RELATIVE accuracy only from a
            // triple whose e makes it as large as the others: a first triple at e = 1 gives the scales, then K is taken at
            // e_K ~ sqrt(|M| / |K|), D at e_D ~ min(|M| / |D|, e_K) and M at min(1, e_K, e_D) (powers of two: the scalings are exact; a triple
            // whose e is 1 again is not repeated).  At e = 1 alone D and K would carry an ABSOLUTE error of ~10 eps |M| per entry.
            struct Triple { double e = 0.0; std::vector<double> H, H2, Hh; };
            auto run = [&](Triple& t, const double e, const std::vector<double>* have) -> int {
                t.e = e;
                if (have) t.H = *have;
                else { t.H.resize(nn); if (int rc = eval_at(e, nullptr, t.H)) return rc; }
                t.H2.resize(nn);
                t.Hh.resize(nn);
                if (int rc = eval_at(2.0 * e, nullptr, t.H2)) return rc;
                return eval_at(0.5 * e, nullptr, t.Hh);
            };
            auto k_of = [](const Triple& t, size_t i) { return (3.0 * t.H[i] - t.H2[i] - 2.0 * t.Hh[i]) * (2.0 / 3.0) / (t.e * t.e); };
            auto d_of = [](const Triple& t, size_t i, double k) { return ((t.H[i] - t.H2[i]) - 3.0 * t.e * t.e * k) / t.e; };
            Triple t1, tK, tD;
            if (int rc = run(t1, 1.0, &H1)) return rc;
            double nM = 0.0, nD = 0.0, nK = 0.0;
            for (size_t i = 0; i < nn; ++i) {
                const double k = k_of(t1, i), d = d_of(t1, i, k), mm = H1[i] + d + k;
                nM += mm * mm; nD += d * d; nK += k * k;
            }
            auto pow2 = [](double x) { return std::ldexp(1.0, (int)std::lround(std::log2(std::min(std::max(x, 0x1p-20), 0x1p20)))); };
            const double eK = nK > 0.0 ? pow2(std::sqrt(std::sqrt(nM / nK))) : 1.0;      // (nM, nD, nK are squared norms)
            const double eD = nD > 0.0 ? std::min(pow2(std::sqrt(nM / nD)), eK) : 1.0;      // (beyond e_K the e |K| term of the roundoff grows)
            const double eM = std::min(1.0, std::min(eK, eD));
            const Triple* pK = &t1;
            if (eK != 1.0) { if (int rc = run(tK, eK, nullptr)) return rc; pK = &tK; }
            const Triple* pD = eD == 1.0 ? &t1 : (eD == eK ? pK : &tD);
            if (pD == &tD) { if (int rc = run(tD, eD, nullptr)) return rc; }
            const std::vector<double>* HM = eM == 1.0 ? &H1 : (eM == eK ? &pK->H : &pD->H);      // (eM is 1, eK or eD)
            for (size_t i = 0; i < nn; ++i) {
                const double k = k_of(*pK, i), d = d_of(*pD, i, k);
                M[i] = (*HM)[i] + eM * d + eM * eM * k;
                D[i] = d;
                if (K) K[i] = k;
            }
        }
    }
    if (dMv) {
        std::vector<double> Hv(nn);
        if (int rc = eval_at(1.0, v, Hv)) return rc;
        for (size_t i = 0; i < nn; ++i) dMv[i] = Hv[i] - H1[i];
    }
    if (dMdq) {
        std::vector<double> Hk(nn), ek(nv);
        for (size_t k = 0; k < nr; ++k) {
            for (size_t i = 0; i < nv; ++i) ek[i] = (i % nr == k) ? 1.0 : 0.0;
            if (int rc = eval_at(1.0, ek.data(), Hk)) return rc;
            for (size_t t = 0; t < (size_t)b->B; ++t)
                for (size_t i = 0; i < nr; ++i)
                    for (size_t r = 0; r < nr; ++r)      // dMdq(r, k, i) = (dMdq(:,:,i) e_k)(r) = column i of H(e_k) - H(0)
                        dMdq[t * nr * nr * nr + r + nr * (k + nr * i)] = Hk[t * nr * nr + i * nr + r] - H1[t * nr * nr + i * nr + r];
        }
    }
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
    d.comp = o->compensated ? 1.0 : 0.0;
    if (o->ls_fail_limit < 0) return fail(RMX_E_INVALID, "opts.ls_fail_limit must be >= 0");
    d.lsFailLimit = o->ls_fail_limit;
    d.parkHalv = 0;
    d.coopTicks = b->m->coop_ticks;
    return RMX_OK;
}

static int launch_step(rmx_batch* b, const rmx_opts* opts, int nsteps, int integ, bool with_stats, double* dT, double* dV,
                       double* dQ = nullptr, double* dQd = nullptr, int* dC = nullptr) {
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
    a.histQ = dQ; a.histQd = dQd; a.histC = dC;
    a.chart = b->chart;
    a.resume = b->resume;
    a.ticks = b->ticks;
    rc = pending_error_check(b, "rmx_step");
    if (rc) return rc;
    // Park and relaunch (rmx_device.h CoopCtx): serial chains of <= 32 nodes with ForceGroundCuboid.  A rollout whose Newton solve keeps
    // running out its line searches is parked by the launch with the contact terms and finished by groups of COOP_G wavefronts that
    // evaluate the reference's trial points side by side - same decisions, same results, the launch no longer waits for one wavefront
    // walking through 20 trials 320 times.  RMX_PARK_HALVINGS=0 switches it off (one wavefront per rollout throughout).
    o.parkHalv = 0;
    m->pair32 = !m->big && m->NP == 32 && m->dm.con && m->dm.is_chain && m->dm.nsph == 0;
    if (m->pair32 && m->n_simd >= COOP_G && b->B >= 1) {
        const char* e = getenv("RMX_PARK_HALVINGS");       // (read at every call: tests and tools switch it inside one process)
        o.parkHalv = e ? atoi(e) : 24;
    }
    if (o.parkHalv > 0) {
        // buffers for the most groups a launch of this batch can ever hold (it alone on the device); each under its own check, so that a
        // failed allocation leaves nothing half set up for the next call
        const int cap = std::min(b->B, m->n_simd / COOP_G);
        if (!b->park) HIPCHK(hipMalloc((void**)&b->park, sizeof(int) * (2 + 4 * (size_t)b->B)));
        if (!b->xch) HIPCHK(hipMalloc((void**)&b->xch, sizeof(unsigned) * COOP_WORDS * (size_t)cap));
        if (!b->xrec) HIPCHK(hipMalloc((void**)&b->xrec, sizeof(unsigned long long) * 2 * COOP_REC * (size_t)cap));
        {
            std::lock_guard<std::mutex> lk(g_coop_mu);
            if (std::find(g_coop_batches.begin(), g_coop_batches.end(), b) == g_coop_batches.end()) g_coop_batches.push_back(b);
        }
        // the groups of this launch: all resident at once (one 512-register wavefront per SIMD) beside those of every other live batch that
        // may be stepping on this device at the same time
        b->ngroups = std::max(1, std::min(cap, (m->n_simd / COOP_G) / std::max(1, coop_batches_on_device(m->device))));
        HIPCHK(hipMemsetAsync(b->park, 0, sizeof(int) * (2 + 4 * (size_t)b->B), b->stream));
        HIPCHK(hipMemsetAsync(b->xch, 0, sizeof(unsigned) * COOP_WORDS * (size_t)b->ngroups, b->stream));
        HIPCHK(hipMemsetAsync(b->xrec, 0, sizeof(unsigned long long) * 2 * COOP_REC * (size_t)b->ngroups, b->stream));
        a.park = b->park;
        a.xch = b->xch;
        a.xrec = b->xrec;
        a.ngroups = b->ngroups;
        const char* cm = getenv("RMX_COOP_MAP");
        a.coop_map = cm ? atoi(cm) : 0;
    }
    {
        const char* ra = getenv("RMX_W2_RUNAHEAD");
        a.w2_noahead = (ra && atoi(ra) == 0) ? 1 : 0;
        const char* pc = getenv("RMX_PAIRC");        // (read at every call, like the others: tests switch it inside one process)
        a.pairc = (pc && atoi(pc) == 0) ? 0 : 1;
    }
    if (m->pair32) {
        // RMX_GROUND_FUSED: 1 (default) ONE launch for the whole call - the rollouts (free flight, then the steps with the contact terms)
        // and, behind them in dispatch order, the cooperative groups that pick the parked rollouts up as they appear: no launch
        // boundary holds a rollout back (rmx_kernels.hip k_ground32); 2 the groups in a second launch; 0 three launches (lean, contact
        // terms, groups); 3 measurement aid (the groups as a second launch of k_ground32)
        const char* f = getenv("RMX_GROUND_FUSED");
        a.fused = f ? atoi(f) : 1;
        if (a.fused == 1 && o.parkHalv <= 0) a.fused = 2;
        if (a.fused && !b->gargs) HIPCHK(hipMalloc(&b->gargs, RMX_GARGS_BYTES));
    }
    // the per-rollout tick counters: the kernels of a call ADD their share (a contact-capable call is up to three launches); the
    // headline kernel (one launch, rmx_kernels.hip RMX_PART 7 - the condition is launch_step_np_32's) stores its count instead, and the
    // fill dispatch ahead of a 0.8 ms launch is saved
    const bool stores_ticks = !m->big && m->NP == 32 && !m->dm.con && m->dm.nsph == 0 && m->dm.is_chain && m->dm.n == 32 && integ == INTEG_BDF1 && a.pairc;
    if (!stores_ticks) HIPCHK(hipMemsetAsync(b->ticks, 0, sizeof(unsigned long long) * b->B, b->stream));
    HIPCHK(hipEventRecord(b->ev0, b->stream));
    if (m->big) launch_big_step(m, b, integ, o, a);
    else DISPATCH_NP(m->NP, launch_step_np, m, b, integ, o, a);
    // BDF2 keeps (q, qdot) of step k-1 in qp/qdp.  BDF1 steps do not maintain them (and, with JointSpherical, may leave q in
    // another Euler chart than qp), so a BDF1 call invalidates the multistep history: the next rmx_step_bdf2 restarts with
    // SDIRK2, as a fresh driverRedMaxBDF2 run from that state would (driverRedMaxBDF2.m:64-88).
    HIPCHK(set_started(b, integ == INTEG_BDF2 ? 1 : 0));   // BDF2: any non-zero value
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(b->ev1, b->stream));
    return RMX_OK;
}

// ---- the per-step record of a step call (Scene.saveHistory, Scene.m:134-161), in device buffers that live on the batch
static void hist_release(rmx_batch* b) { b->hist = rmx_batch::Hist{}; }      // forget the record, keep the buffers
static void hist_free(rmx_batch* b) {                                          // batch teardown
    for (void* p : {(void*)b->hpool.T, (void*)b->hpool.V, (void*)b->hpool.Q, (void*)b->hpool.Qd, (void*)b->hpool.C})
        if (p) (void)hipFree(p);
    b->hpool = rmx_batch::HistPool{};
    hist_release(b);
}
template <class T>
static hipError_t pool_grow(rmx_batch* b, T*& p, size_t& cap, size_t need, T** second = nullptr) {
    if (need <= cap) return hipSuccess;
    // growing frees the old buffer: whatever this batch still has in flight may be writing it
    hipError_t e = hipStreamSynchronize(b->stream);
    if (e != hipSuccess) return e;
    if (p) (void)hipFree(p);
    p = nullptr;
    if (second && *second) { (void)hipFree(*second); *second = nullptr; }
    cap = 0;
    e = hipMalloc((void**)&p, need * sizeof(T));
    if (e == hipSuccess && second) e = hipMalloc((void**)second, need * sizeof(T));
    if (e == hipSuccess) cap = need;
    return e;
}
static int hist_alloc(rmx_batch* b, int nsteps, int record) {
    hist_release(b);
    const rmx_model* m = b->m;
    const size_t nh = (size_t)nsteps * b->B, nq = nh * m->nr, nc = nh * (size_t)m->dm.nsph;
    hipError_t e = hipSuccess;
    if ((record & RMX_REC_ENERGY) && nh) {
        e = pool_grow(b, b->hpool.T, b->hpool.capH, nh, &b->hpool.V);
        if (e == hipSuccess) { b->hist.T = b->hpool.T; b->hist.V = b->hpool.V; }
    }
    if (e == hipSuccess && (record & RMX_REC_STATE) && nq) {
        e = pool_grow(b, b->hpool.Q, b->hpool.capQ, nq, &b->hpool.Qd);
        if (e == hipSuccess) { b->hist.Q = b->hpool.Q; b->hist.Qd = b->hpool.Qd; }
    }
    // models with spherical joints run the extended (CT) step kernels, which record the chart after every step
    if (e == hipSuccess && (record & RMX_REC_CHARTS) && nc) {
        e = pool_grow<int>(b, b->hpool.C, b->hpool.capC, nc);
        if (e == hipSuccess) b->hist.C = b->hpool.C;
    }
    if (e != hipSuccess) {
        hist_free(b);
        (void)hipGetLastError();      // the failed allocation must not be taken for a failed launch later on
        return fail(RMX_E_NOMEM, std::string("hipMalloc(per-step record): ") + hipGetErrorString(e));
    }
    b->hist.nsteps = nsteps;
    return RMX_OK;
}
// rows of `width` bytes, `rows` of them: device (dense) -> host array whose rows are `dpitch` bytes apart
static hipError_t rows_out(void* dst, size_t dpitch, const void* src, size_t width, size_t rows, hipStream_t st) {
    if (!width || !rows) return hipSuccess;
    if (dpitch == width) return hipMemcpyAsync(dst, src, width * rows, hipMemcpyDeviceToHost, st);
    return hipMemcpy2DAsync(dst, dpitch, src, width, width, rows, hipMemcpyDeviceToHost, st);
}
// Enqueue the copies of the record into host arrays laid out for `pitchB` trajectories per step, this batch's first trajectory
// being number `first` of them (pitchB = B, first = 0 for a batch on its own; a shard of an rmx_group otherwise).
static int hist_copy_out(rmx_batch* b, const rmx_history* h, size_t pitchB, size_t first) {
    if (!h) return RMX_OK;
    const rmx_model* m = b->m;
    const size_t K = (size_t)b->hist.nsteps, B = (size_t)b->B, nr = (size_t)m->nr, ns = (size_t)m->dm.nsph;
    if ((h->T == nullptr) != (h->V == nullptr)) return fail(RMX_E_INVALID, "history T and V must be given together");
    if ((h->q == nullptr) != (h->qdot == nullptr)) return fail(RMX_E_INVALID, "history q and qdot must be given together");
    if (K && B && ((h->T && !b->hist.T) || (h->q && nr && !b->hist.Q) || (h->charts && ns && !b->hist.C)))
        return fail(RMX_E_INVALID, "this part of the per-step record was not recorded by the step call (RMX_REC_*)");
    hipError_t e = hipSuccess;
    if (h->T) {
        e = rows_out(h->T + first, pitchB * 8, b->hist.T, B * 8, K, b->stream);
        if (e == hipSuccess) e = rows_out(h->V + first, pitchB * 8, b->hist.V, B * 8, K, b->stream);
    }
    if (e == hipSuccess && h->q && nr) {
        e = rows_out(h->q + first * nr, pitchB * nr * 8, b->hist.Q, B * nr * 8, K, b->stream);
        if (e == hipSuccess) e = rows_out(h->qdot + first * nr, pitchB * nr * 8, b->hist.Qd, B * nr * 8, K, b->stream);
    }
    if (e == hipSuccess && h->charts && ns) e = rows_out(h->charts + first * ns, pitchB * ns * 4, b->hist.C, B * ns * 4, K, b->stream);
    if (e != hipSuccess) return fail(RMX_E_HIP, std::string("copying the per-step record: ") + hipGetErrorString(e));
    return RMX_OK;
}
static hipError_t stats_copy_out(rmx_batch* b, const rmx_stats* st) {
    hipError_t e = hipSuccess;
    if (!st) return e;
    if (st->newton_iters) e = hipMemcpyAsync(st->newton_iters, b->it, sizeof(int) * b->B, hipMemcpyDeviceToHost, b->stream);
    if (e == hipSuccess && st->ls_halvings) e = hipMemcpyAsync(st->ls_halvings, b->ls, sizeof(int) * b->B, hipMemcpyDeviceToHost, b->stream);
    if (e == hipSuccess && st->status) e = hipMemcpyAsync(st->status, b->status, sizeof(int) * b->B, hipMemcpyDeviceToHost, b->stream);
    return e;
}
// the device flag that tells BDF2 whether (q, qdot) of step k-1 are in place: written only when its value changes (one fill dispatch
// less on the stream of every step call: ~5 us of a 0.8 ms launch at the driver's --steps 20)
static hipError_t set_started(rmx_batch* b, const int v) {
    if (b->started_host == v) return hipSuccess;
    const hipError_t e = hipMemsetAsync(b->started, v, sizeof(int), b->stream);
    b->started_host = e == hipSuccess ? v : -1;
    return e;
}
// Wait for a stream by polling (hipStreamSynchronize's wake-up costs ~10 us: 1 - 2 % of the 0.8 ms launches of configs[1] and [3]);
// after RMX_WAIT_SPIN_MS milliseconds (default 2; polling through the 5.8 / 16.6 ms launches of configs[2] / [4] gained nothing
// measurable) the host thread blocks as before.
static hipError_t wait_stream_short(hipStream_t stream) {
    static const int spin_ms = [] { const char* e = getenv("RMX_WAIT_SPIN_MS"); const int v = e ? atoi(e) : 2; return v < 0 ? 0 : v; }();
    hipError_t es = hipErrorNotReady;
    for (const auto t0 = std::chrono::steady_clock::now(); std::chrono::steady_clock::now() - t0 < std::chrono::milliseconds(spin_ms);) {
        es = hipStreamQuery(stream);
        if (es != hipErrorNotReady) break;
    }
    if (es == hipErrorNotReady) es = hipStreamSynchronize(stream);
    else (void)hipGetLastError();      // (hipStreamQuery leaves hipErrorNotReady as the thread's last error while it polls)
    return es;
}

static void take_event_time(rmx_batch* b) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, b->ev0, b->ev1) == hipSuccess) b->last_ms = ms;
}

static int step_sync(rmx_batch* b, const rmx_opts* opts, int nsteps, rmx_stats* st, double* hT, double* hV, int integ,
                     double* hQ = nullptr, double* hQd = nullptr, int* hC = nullptr) {
    if (!b) return fail(RMX_E_INVALID, "null batch");
    if (nsteps < 0) return fail(RMX_E_INVALID, "nsteps < 0");
    if ((hT == nullptr) != (hV == nullptr)) return fail(RMX_E_INVALID, "hist_T and hist_V must be given together");
    if ((hQ == nullptr) != (hQd == nullptr)) return fail(RMX_E_INVALID, "history q and qdot must be given together");
    rmx_model* m = b->m;
    HIPCHK(hipSetDevice(m->device));
    if (nsteps == 0 || m->nr == 0) return RMX_OK;
    if (b->async_pending && (b->hist.T || b->hist.Q || b->hist.C))
        // a recorded *_async launch is still in flight and its record has not been read: a synchronous call would replace it
        // (include/redmax_hip.h "Asynchronous stepping": only *_async / rmx_sync / rmx_stats_reset until rmx_sync)
        return fail(RMX_E_INVALID, "a recorded asynchronous step is pending on this batch: rmx_sync / rmx_history_read first");
    int rc = hist_alloc(b, nsteps, (hT ? RMX_REC_ENERGY : 0) | (hQ ? RMX_REC_STATE : 0) | (hC ? RMX_REC_CHARTS : 0));
    if (rc) return rc;
    const bool ws = st != nullptr;
    if (ws) {
        (void)hipMemsetAsync(b->it, 0, sizeof(int) * b->B, b->stream);
        (void)hipMemsetAsync(b->ls, 0, sizeof(int) * b->B, b->stream);
        (void)hipMemsetAsync(b->status, 0, sizeof(int) * b->B, b->stream);
    }
    rc = launch_step(b, opts, nsteps, integ, ws, b->hist.T, b->hist.V, b->hist.Q, b->hist.Qd, b->hist.C);
    hipError_t e = hipSuccess;
    if (rc == RMX_OK) {
        const rmx_history h{hT, hV, hQ, hQd, hC};
        rc = hist_copy_out(b, &h, (size_t)b->B, 0);
        if (rc == RMX_OK) {
            if (ws) e = stats_copy_out(b, st);
            if (e == hipSuccess) e = wait_stream_short(b->stream);
            if (e == hipSuccess) {
                take_event_time(b);
                b->async_pending = false;      // the stream has been waited for: nothing of this batch is in flight any more
            }
        }
    }
    hist_release(b);                            // a synchronous call has delivered its record: nothing to keep (the buffers stay)
    if (rc) return rc;
    if (e != hipSuccess) return fail(RMX_E_HIP, std::string("rmx_step: ") + hipGetErrorString(e));
    return RMX_OK;
}

// Scene.saveHistory (Scene.m:134-161) for every step and trajectory: t is k*h, the rest comes back here.
extern "C" int rmx_step_history(rmx_batch* b, const rmx_opts* opts, int nsteps, int integrator, rmx_stats* stats, const rmx_history* hist) {
    if (integrator != 1 && integrator != 2) return fail(RMX_E_INVALID, "integrator must be 1 (BDF1) or 2 (BDF2)");
    if (!hist) return fail(RMX_E_INVALID, "null history");
    return step_sync(b, opts, nsteps, stats, hist->T, hist->V, integrator == 1 ? INTEG_BDF1 : INTEG_BDF2, hist->q, hist->qdot, hist->charts);
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
    if (b->m->dm.nsph) return fail(RMX_E_INVALID, "rmx_step_euler: matlab-simple has no JointSpherical; use rmx_step_bdf1/bdf2");
    if (b->m->big) return fail(RMX_E_INVALID, "rmx_step_euler: trees of more than 64 nodes run rmx_step_bdf1/bdf2 only");
    if (nsteps < 0 || !(h > 0)) return fail(RMX_E_INVALID, "bad nsteps / h");
    if ((hT == nullptr) != (hV == nullptr)) return fail(RMX_E_INVALID, "hist_T and hist_V must be given together");
    rmx_model* m = b->m;
    HIPCHK(hipSetDevice(m->device));
    if (nsteps == 0 || m->nr == 0) return RMX_OK;
    if (int rc = pending_error_check(b, "rmx_step_euler")) return rc;
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
    if (e == hipSuccess) e = set_started(b, 0);
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


static int adjoint_impl(rmx_batch* b, const rmx_opts* opts, int nsteps, const rmx_task_pointpos* task, const double* p, double* P,
                        double* dPdp, rmx_stats* stats, const int integ, const bool on_device = false) {
    if (!b || !task || !p || !P || !dPdp) return fail(RMX_E_INVALID, "null argument");
    rmx_model* m = b->m;
    if (nsteps < 1) return fail(RMX_E_INVALID, "nsteps < 1");
    if (m->dm.con) return fail(RMX_E_INVALID, "rmx_adjoint: ground contact is outside the adjoint path (SURVEY.md 8(f))");
    if (m->dm.nsph) return fail(RMX_E_INVALID, "rmx_adjoint: spherical joints are outside the adjoint path (SURVEY.md 8(f))");
    if (m->big) return fail(RMX_E_INVALID, "rmx_adjoint: trees of more than 64 nodes are outside the adjoint path");
    if (task->body < 0 || task->body >= m->nlist) return fail(RMX_E_INVALID, "task body out of range");
    if (task->step < 1 || task->step > nsteps) return fail(RMX_E_INVALID, "task step must be in [1, nsteps]");
    HIPCHK(hipSetDevice(m->device));
    DevOpts o;
    int rc = make_opts(b, opts, o);
    if (rc) return rc;
    rc = pending_error_check(b, "rmx_adjoint");
    if (rc) return rc;
    const size_t nn = (size_t)m->n * m->n, nv = (size_t)b->B * m->nr;
    const size_t hist = (size_t)b->B * nsteps * nn * sizeof(double);
    if (3 * hist > ((size_t)200 << 30)) return fail(RMX_E_NOMEM, "adjoint history (H, M, D per step) would exceed 200 GiB");
    AdjArgs a{};
    a.B = b->B; a.nsteps = nsteps; a.task_step = task->step; a.task_node = m->node_of_listing[task->body];
    for (int c = 0; c < 3; ++c) { a.xl[c] = task->xlocal[c]; a.xt[c] = task->xtarget[c]; }
    a.pscale = task->pscale; a.wreg = task->wreg; a.wpos = task->wpos;
    a.q = b->q; a.qd = b->qd; a.qp = b->qp; a.qdp = b->qdp; a.p = b->tmpA;
    a.it = stats ? b->it : nullptr; a.status = b->status;
    void* bufs[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    const size_t sizes[6] = {hist, hist, hist, (size_t)b->B * m->n * sizeof(double), (size_t)b->B * sizeof(double), nv * sizeof(double)};
    hipError_t e = hipSuccess;
    {   // one workspace per batch, kept between calls (grow-only): parts at 256-byte boundaries
        size_t total = 0, offs[6];
        for (int i = 0; i < 6; ++i) {
            offs[i] = total;
            total += (sizes[i] + 255) & ~(size_t)255;
        }
        if (total > b->adjws_bytes) {
            void* fresh = nullptr;
            e = hipMalloc(&fresh, total);      // the new buffer first: a failed regrow keeps the old workspace usable
            if (e != hipSuccess && b->adjws) {      // ... unless old + new do not fit side by side: then the old one has to go first
                (void)hipGetLastError();
                (void)hipStreamSynchronize(b->stream);
                (void)hipFree(b->adjws);
                b->adjws = nullptr;
                b->adjws_bytes = 0;
                e = hipMalloc(&fresh, total);
            }
            if (e == hipSuccess) {
                if (b->adjws) { (void)hipStreamSynchronize(b->stream); (void)hipFree(b->adjws); }
                b->adjws = fresh;
                b->adjws_bytes = total;
            } else {
                (void)hipGetLastError();       // an out-of-memory here must not be taken for a failed launch by the next call
            }
        }
        if (e == hipSuccess)
            for (int i = 0; i < 6; ++i) bufs[i] = (char*)b->adjws + offs[i];
    }
    // (dPdq needs no fill: the task step lies in [1, nsteps] - checked above -, so the forward kernel writes every entry the backward
    // kernel reads; one dispatch less ahead of a 0.83 ms launch pair)
    if (e == hipSuccess && !on_device) e = hipMemcpyAsync(b->tmpA, p, nv * sizeof(double), hipMemcpyHostToDevice, b->stream);
    if (e == hipSuccess) {
        a.Hs = (double*)bufs[0]; a.Ms = (double*)bufs[1]; a.Ds = (double*)bufs[2];
        a.dPdq = (double*)bufs[3]; a.P = (double*)bufs[4]; a.dPdp = (double*)bufs[5];
        if (on_device) {          // the caller's device arrays directly: nothing staged, nothing copied back
            a.p = p; a.P = P; a.dPdp = dPdp;
        }
        e = hipEventRecord(b->ev0, b->stream);
            DISPATCH_NP(m->NP, launch_adjoint, m, b, integ, o, a);
        if (e == hipSuccess) e = hipGetLastError();
        if (e == hipSuccess) e = hipEventRecord(b->ev1, b->stream);
        // after a BDF2 rollout (q, qdot) of step k-1 are in place: rmx_step_bdf2 may continue it; a BDF1 rollout invalidates them
        if (e == hipSuccess) e = set_started(b, integ == INTEG_BDF2 ? 1 : 0);
        if (e == hipSuccess && !on_device) e = hipMemcpyAsync(P, a.P, sizes[4], hipMemcpyDeviceToHost, b->stream);
        if (e == hipSuccess && !on_device) e = hipMemcpyAsync(dPdp, a.dPdp, sizes[5], hipMemcpyDeviceToHost, b->stream);
        if (e == hipSuccess && stats) {
            if (stats->newton_iters) e = hipMemcpyAsync(stats->newton_iters, b->it, sizeof(int) * b->B, hipMemcpyDeviceToHost, b->stream);
            if (e == hipSuccess && stats->status) e = hipMemcpyAsync(stats->status, b->status, sizeof(int) * b->B, hipMemcpyDeviceToHost, b->stream);
        }
        if (e == hipSuccess) e = wait_stream_short(b->stream);      // (configs[3]: a launch pair of 0.9 ms)
        if (e == hipSuccess) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, b->ev0, b->ev1) == hipSuccess) b->last_ms = ms;
        }
    }
    if (e == hipErrorOutOfMemory) return fail(RMX_E_NOMEM, "rmx_adjoint: no device memory for the history (H, M, D per step and rollout)");
    if (e != hipSuccess) return fail(RMX_E_HIP, std::string("rmx_adjoint: ") + hipGetErrorString(e));
    return RMX_OK;
}

extern "C" int rmx_adjoint_bdf1(rmx_batch* b, const rmx_opts* opts, int nsteps, const rmx_task_pointpos* task, const double* p,
                                double* P, double* dPdp, rmx_stats* stats) {
    return adjoint_impl(b, opts, nsteps, task, p, P, dPdp, stats, INTEG_BDF1);
}
extern "C" int rmx_adjoint_bdf2(rmx_batch* b, const rmx_opts* opts, int nsteps, const rmx_task_pointpos* task, const double* p,
                                double* P, double* dPdp, rmx_stats* stats) {
    return adjoint_impl(b, opts, nsteps, task, p, P, dPdp, stats, INTEG_BDF2);
}

extern "C" int rmx_adjoint_bdf1_device(rmx_batch* b, const rmx_opts* opts, int nsteps, const rmx_task_pointpos* task, const double* d_p,
                                       double* d_P, double* d_dPdp, rmx_stats* stats) {
    return adjoint_impl(b, opts, nsteps, task, d_p, d_P, d_dPdp, stats, INTEG_BDF1, true);
}
extern "C" int rmx_adjoint_bdf2_device(rmx_batch* b, const rmx_opts* opts, int nsteps, const rmx_task_pointpos* task, const double* d_p,
                                       double* d_P, double* d_dPdp, rmx_stats* stats) {
    return adjoint_impl(b, opts, nsteps, task, d_p, d_P, d_dPdp, stats, INTEG_BDF2, true);
}

// simLoop of one batch enqueued on its stream, nothing waited for (include/redmax_hip.h "Asynchronous stepping")
static int step_async(rmx_batch* b, const rmx_opts* opts, int nsteps, int integ, int record) {
    if (!b) return fail(RMX_E_INVALID, "null batch");
    if (nsteps < 0) return fail(RMX_E_INVALID, "nsteps < 0");
    if (record & ~(RMX_REC_ENERGY | RMX_REC_STATE | RMX_REC_CHARTS)) return fail(RMX_E_INVALID, "record: unknown RMX_REC_* bits");
    HIPCHK(hipSetDevice(b->m->device));
    if (b->async_pending && (b->hist.T || b->hist.Q || b->hist.C)) {
        // the launch in flight writes the record that is about to be replaced: wait for it (stacked async steps without a record
        // need no wait - the stream orders them)
        HIPCHK(hipStreamSynchronize(b->stream));
    }
    int rc = hist_alloc(b, nsteps, record);
    if (rc) return rc;
    if (nsteps == 0 || b->m->nr == 0) return RMX_OK;
    rc = launch_step(b, opts, nsteps, integ, true, b->hist.T, b->hist.V, b->hist.Q, b->hist.Qd, b->hist.C);   // counters accumulate on the device
    if (rc == RMX_OK) b->async_pending = true;      // until rmx_sync (or any synchronous call on this batch) has waited for it
    return rc;
}
extern "C" int rmx_step_bdf1_async(rmx_batch* b, const rmx_opts* opts, int nsteps) { return step_async(b, opts, nsteps, INTEG_BDF1, 0); }
extern "C" int rmx_step_bdf2_async(rmx_batch* b, const rmx_opts* opts, int nsteps) { return step_async(b, opts, nsteps, INTEG_BDF2, 0); }
extern "C" int rmx_step_history_async(rmx_batch* b, const rmx_opts* opts, int nsteps, int integrator, int record) {
    if (integrator != 1 && integrator != 2) return fail(RMX_E_INVALID, "integrator must be 1 (BDF1) or 2 (BDF2)");
    return step_async(b, opts, nsteps, integrator == 1 ? INTEG_BDF1 : INTEG_BDF2, record);
}
static int history_read(rmx_batch* b, const rmx_history* hist, size_t pitchB, size_t first) {
    if (!b || !hist) return fail(RMX_E_INVALID, "null argument");
    HIPCHK(hipSetDevice(b->m->device));
    if (b->async_pending) {
        const int rc = rmx_sync(b);
        if (rc) return rc;
    }
    const int rc = hist_copy_out(b, hist, pitchB, first);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(b->stream));
    return RMX_OK;
}
extern "C" int rmx_history_read(rmx_batch* b, const rmx_history* hist) { return history_read(b, hist, b ? (size_t)b->B : 0, 0); }
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
    b->async_pending = false;
    return RMX_OK;
}
// How the time of the last rmx_step_* launch was spread over the rollouts: shader-clock ticks (s_memtime) every rollout's wavefront
// spent inside the kernel(s).  A launch ends with its slowest rollout; this is the per-rollout distribution behind that.
extern "C" int rmx_step_ticks(rmx_batch* b, unsigned long long* ticks) {
    if (!b || !ticks) return fail(RMX_E_INVALID, "null argument");
    HIPCHK(hipSetDevice(b->m->device));
    HIPCHK(hipMemcpyAsync(ticks, b->ticks, sizeof(unsigned long long) * b->B, hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    b->async_pending = false;
    return RMX_OK;
}

extern "C" int rmx_sync(rmx_batch* b) {
    if (!b) return fail(RMX_E_INVALID, "null batch");
    HIPCHK(hipSetDevice(b->m->device));
    const hipError_t es = wait_stream_short(b->stream);
    b->async_pending = false;
    if (es != hipSuccess) return fail(RMX_E_HIP, std::string("rmx_sync: the asynchronous launch failed: ") + hipGetErrorString(es));
    take_event_time(b);
    return RMX_OK;
}

extern "C" int rmx_profile_phases(rmx_batch* b, int reps, double h, double* cycles4) {
    if (!b || !cycles4 || reps < 1) return fail(RMX_E_INVALID, "bad argument");
    rmx_model* m = b->m;
    if (m->big) return fail(RMX_E_INVALID, "rmx_profile_phases: one-wavefront kernels only (trees of up to 64 nodes)");
    HIPCHK(hipSetDevice(m->device));
    if (int rc = pending_error_check(b, "rmx_profile_phases")) return rc;
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
    if (int rc = pending_error_check(b, "rmx_energy")) return rc;
    double *dT = nullptr, *dV = nullptr;
    HIPCHK(hipMalloc((void**)&dT, sizeof(double) * b->B));
    hipError_t e = hipMalloc((void**)&dV, sizeof(double) * b->B);
    if (e != hipSuccess) { (void)hipFree(dT); return fail(RMX_E_NOMEM, "hipMalloc(energy)"); }
    if (m->big) launch_big_energy(m, b, dT, dV);
    else DISPATCH_NP(m->NP, launch_energy, m, b, dT, dV);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(T, dT, sizeof(double) * b->B, hipMemcpyDeviceToHost, b->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(V, dV, sizeof(double) * b->B, hipMemcpyDeviceToHost, b->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
    (void)hipFree(dT);
    (void)hipFree(dV);
    if (e != hipSuccess) return fail(RMX_E_HIP, std::string("rmx_energy: ") + hipGetErrorString(e));
    return RMX_OK;
}


// ============================================================================ multi-device groups (include/redmax_hip.h, ABI 107)
#include <chrono>

// ---- RCCL, bound at run time.  libredmax_hip.so links the HIP runtime only; the one collective of the path (north_star: "RCCL over
// xGMI only for the final gather") needs librccl when a group spans several DISTINCT devices, and a host without it still simulates
// (rmx_group_gather_device then says so).  The entry points are resolved from librccl.so(.1) on first use; the types come from rccl.h.
#include <dlfcn.h>
#include <rccl/rccl.h>
namespace {
struct Rccl {
    void* lib = nullptr;
    bool tried = false;
    std::string why;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mu;
bool rccl_load() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.tried) return g_rccl.lib != nullptr;
    g_rccl.tried = true;
    const char* env = getenv("RMX_RCCL_LIB");
    const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* nm : names) {
        if (!nm || !*nm) continue;
        g_rccl.lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
        if (g_rccl.lib) break;
        g_rccl.why = dlerror() ? dlerror() : "dlopen failed";
    }
    if (!g_rccl.lib) return false;
    bool ok = true;
#define RMX_SYM(field, name) ok = ok && ((g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(g_rccl.lib, name))) != nullptr)
    RMX_SYM(CommInitAll, "ncclCommInitAll"); RMX_SYM(CommDestroy, "ncclCommDestroy"); RMX_SYM(GetErrorString, "ncclGetErrorString");
    RMX_SYM(AllGather, "ncclAllGather"); RMX_SYM(Broadcast, "ncclBroadcast"); RMX_SYM(Send, "ncclSend"); RMX_SYM(Recv, "ncclRecv");
    RMX_SYM(GroupStart, "ncclGroupStart"); RMX_SYM(GroupEnd, "ncclGroupEnd");
#undef RMX_SYM
    if (!ok) {
        g_rccl.why = "librccl lacks an entry point this library binds";
        dlclose(g_rccl.lib);
        g_rccl.lib = nullptr;
    }
    return ok;
}
}  // namespace

struct rmx_group {
    int B = 0, nr = 0;
    std::vector<rmx_model*> models;
    std::vector<rmx_batch*> batches;
    std::vector<int> device, first, count;
    std::chrono::steady_clock::time_point t0;
    double wall_ms = 0.0;
    bool in_flight = false;
    // rmx_group_gather_device: one RCCL communicator per shard (a single-process clique, ncclCommInitAll on the device list), created
    // at the first gather of a group whose devices are pairwise distinct; a group that lists a device twice gathers by peer copies
    std::vector<ncclComm_t> comms;
    int gather_path = 0;            // 0 not decided, 1 RCCL, 2 peer copies (a device is listed more than once, or RMX_GROUP_GATHER=copy)
    const char* last_gather = "";   // "rccl:allgather" | "rccl:broadcast" | "rccl:sendrecv" | "copy" (rmx_group_gather_path)
    std::vector<double*> gq, gqd;   // rmx_group_gather: the group's own [batch][nr] destinations, one pair per shard's device (on demand)
};

extern "C" void rmx_group_destroy(rmx_group* g) {
    if (!g) return;
    for (ncclComm_t c : g->comms)
        if (c && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c);
    for (size_t s = 0; s < g->gq.size(); ++s) {
        if (g->gq[s] || g->gqd[s]) (void)hipSetDevice(g->device[s]);
        if (g->gq[s]) (void)hipFree(g->gq[s]);
        if (g->gqd[s]) (void)hipFree(g->gqd[s]);
    }
    for (rmx_batch* b : g->batches) rmx_batch_destroy(b);
    for (rmx_model* m : g->models) rmx_model_destroy(m);
    delete g;
}

extern "C" int rmx_group_create(const rmx_model_desc* desc, const rmx_ground_contact* gc, int batch, const int* devices, int ndevices,
                                rmx_group** out) {
    if (!desc || !out) return fail(RMX_E_INVALID, "null argument");
    *out = nullptr;
    if (batch < 1) return fail(RMX_E_INVALID, "batch must be >= 1");
    if (ndevices < 1) return fail(RMX_E_INVALID, "ndevices must be >= 1");
    if (ndevices > batch) return fail(RMX_E_INVALID, "more shards than trajectories");
    rmx_group* g = new rmx_group();
    g->B = batch;
    // contiguous shards in device-list order, sizes differing by at most one (sharding.plan(..., "strong"))
    const int base = batch / ndevices, extra = batch % ndevices;
    int at = 0;
    for (int s = 0; s < ndevices; ++s) {
        const int cnt = base + (s < extra ? 1 : 0);
        rmx_model* m = nullptr;
        rmx_batch* b = nullptr;
        int rc = rmx_model_create(desc, devices ? devices[s] : s, &m);
        if (rc == RMX_OK) {
            g->models.push_back(m);
            if (gc) rc = rmx_model_set_ground_contact(m, gc);
        }
        if (rc == RMX_OK) rc = rmx_batch_create(m, cnt, &b);
        if (rc != RMX_OK) {
            const std::string msg = "rmx_group_create, shard " + std::to_string(s) + ": " + g_err;
            rmx_group_destroy(g);
            return fail(rc, msg);
        }
        g->batches.push_back(b);
        g->device.push_back(m->device);
        g->first.push_back(at);
        g->count.push_back(cnt);
        at += cnt;
    }
    g->nr = g->models[0]->nr;
    *out = g;
    return RMX_OK;
}
extern "C" int rmx_group_batch_size(const rmx_group* g) { return g ? g->B : RMX_E_INVALID; }
extern "C" int rmx_group_nshards(const rmx_group* g) { return g ? (int)g->batches.size() : RMX_E_INVALID; }
extern "C" int rmx_group_shard(const rmx_group* g, int s, int* device, int* first, int* count) {
    if (!g || s < 0 || s >= (int)g->batches.size()) return fail(RMX_E_INVALID, "rmx_group_shard: bad argument");
    if (device) *device = g->device[s];
    if (first) *first = g->first[s];
    if (count) *count = g->count[s];
    return RMX_OK;
}
extern "C" rmx_batch* rmx_group_shard_batch(rmx_group* g, int s) { return (g && s >= 0 && s < (int)g->batches.size()) ? g->batches[s] : nullptr; }
extern "C" rmx_model* rmx_group_shard_model(rmx_group* g, int s) { return (g && s >= 0 && s < (int)g->models.size()) ? g->models[s] : nullptr; }

extern "C" int rmx_group_set_state(rmx_group* g, const double* q, const double* qdot) {
    if (!g) return fail(RMX_E_INVALID, "null group");
    for (size_t s = 0; s < g->batches.size(); ++s) {
        const size_t off = (size_t)g->first[s] * g->nr;
        if (int rc = rmx_set_state(g->batches[s], q ? q + off : nullptr, qdot ? qdot + off : nullptr)) return rc;
    }
    return RMX_OK;
}
extern "C" int rmx_group_get_state(rmx_group* g, double* q, double* qdot) {
    if (!g) return fail(RMX_E_INVALID, "null group");
    for (size_t s = 0; s < g->batches.size(); ++s) {
        const size_t off = (size_t)g->first[s] * g->nr;
        if (int rc = rmx_get_state(g->batches[s], q ? q + off : nullptr, qdot ? qdot + off : nullptr)) return rc;
    }
    return RMX_OK;
}
// The final gather of the path with DEVICE-resident destinations (north_star: "RCCL over xGMI only for the final gather"; SURVEY.md
// 8(e): one collective at the end of the rollout, [B/N][nr] x 2 per shard).  d_q[s], d_qdot[s]: [batch][nr] arrays on shard s's device
// (null for a shard that receives nothing).  root_or_all = RMX_GATHER_ALL: every shard's device ends up with the whole batch
// (ncclAllGather when the shards are equal, one ncclBroadcast per shard inside a group call otherwise); a shard index: that shard's
// device alone (ncclSend / ncclRecv).  The collective is enqueued on the shards' own streams, behind their step kernels - no host
// synchronisation between the rollout and the gather -, and the call returns when every stream has drained.
static int gather_fail(rmx_group* g, ncclResult_t r, const char* what) {
    (void)g;
    return fail(RMX_E_HIP, std::string("rmx_group_gather_device: ") + what + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error"));
}
extern "C" int rmx_group_gather_device(rmx_group* g, double* const* d_q, double* const* d_qdot, int root_or_all) {
    if (!g || !d_q || !d_qdot) return fail(RMX_E_INVALID, "null argument");
    const int S = (int)g->batches.size();
    const bool all = root_or_all == RMX_GATHER_ALL;
    if (!all && (root_or_all < 0 || root_or_all >= S)) return fail(RMX_E_INVALID, "rmx_group_gather_device: root must be a shard index or RMX_GATHER_ALL");
    for (int s = 0; s < S; ++s)
        if ((all || s == root_or_all) && (!d_q[s] || !d_qdot[s])) return fail(RMX_E_INVALID, "rmx_group_gather_device: a receiving shard has no destination");
    if (g->gather_path == 0) {
        bool dup = false;
        for (int a = 0; a < S; ++a)
            for (int c = a + 1; c < S; ++c) dup = dup || g->device[a] == g->device[c];
        const char* force = getenv("RMX_GROUP_GATHER");
        if (dup || (force && !strcmp(force, "copy"))) {
            g->gather_path = 2;       // RCCL refuses a clique that names a GPU twice: such shards exchange by (peer) copies
        } else {
            if (!rccl_load()) return fail(RMX_E_HIP, "rmx_group_gather_device: librccl could not be loaded (" + g_rccl.why + "); RMX_GROUP_GATHER=copy gathers by peer copies");
            g->comms.assign(S, nullptr);
            const ncclResult_t r = g_rccl.CommInitAll(g->comms.data(), S, g->device.data());
            if (r != ncclSuccess) { g->comms.clear(); return gather_fail(g, r, "ncclCommInitAll"); }
            g->gather_path = 1;
        }
    }
    const size_t nr = (size_t)g->nr;
    if (g->gather_path == 2) {
        g->last_gather = "copy";
        for (int src = 0; src < S; ++src) {
            rmx_batch* b = g->batches[src];
            const size_t off = (size_t)g->first[src] * nr, bytes = (size_t)g->count[src] * nr * sizeof(double);
            for (int dst = 0; dst < S; ++dst) {
                if (!(all || dst == root_or_all)) continue;
                // on the SOURCE shard's stream, behind its step kernel; the destination is idle memory of the caller's
                HIPCHK(hipSetDevice(g->device[src]));
                if (g->device[src] == g->device[dst]) {
                    HIPCHK(hipMemcpyAsync(d_q[dst] + off, b->q, bytes, hipMemcpyDeviceToDevice, b->stream));
                    HIPCHK(hipMemcpyAsync(d_qdot[dst] + off, b->qd, bytes, hipMemcpyDeviceToDevice, b->stream));
                } else {
                    HIPCHK(hipMemcpyPeerAsync(d_q[dst] + off, g->device[dst], b->q, g->device[src], bytes, b->stream));
                    HIPCHK(hipMemcpyPeerAsync(d_qdot[dst] + off, g->device[dst], b->qd, g->device[src], bytes, b->stream));
                }
            }
        }
    } else {
        bool equal = true;
        for (int s = 1; s < S; ++s) equal = equal && g->count[s] == g->count[0];
        ncclResult_t r = g_rccl.GroupStart();
        if (r != ncclSuccess) return gather_fail(g, r, "ncclGroupStart");
        if (all && equal) {
            g->last_gather = "rccl:allgather";
            const size_t cnt = (size_t)g->count[0] * nr;
            for (int s = 0; s < S && r == ncclSuccess; ++s) {
                r = g_rccl.AllGather(g->batches[s]->q, d_q[s], cnt, ncclDouble, g->comms[s], g->batches[s]->stream);
                if (r == ncclSuccess) r = g_rccl.AllGather(g->batches[s]->qd, d_qdot[s], cnt, ncclDouble, g->comms[s], g->batches[s]->stream);
            }
        } else if (all) {
            g->last_gather = "rccl:broadcast";      // shards that differ by one rollout: the all-gather as one broadcast per shard
            for (int root = 0; root < S && r == ncclSuccess; ++root) {
                const size_t off = (size_t)g->first[root] * nr, cnt = (size_t)g->count[root] * nr;
                for (int s = 0; s < S && r == ncclSuccess; ++s) {
                    r = g_rccl.Broadcast(g->batches[root]->q, d_q[s] + off, cnt, ncclDouble, root, g->comms[s], g->batches[s]->stream);
                    if (r == ncclSuccess) r = g_rccl.Broadcast(g->batches[root]->qd, d_qdot[s] + off, cnt, ncclDouble, root, g->comms[s], g->batches[s]->stream);
                }
            }
        } else {
            g->last_gather = "rccl:sendrecv";
            const int root = root_or_all;
            for (int s = 0; s < S && r == ncclSuccess; ++s) {
                const size_t off = (size_t)g->first[s] * nr, cnt = (size_t)g->count[s] * nr;
                if (s == root) {      // the root's own slice: a copy on its stream
                    if (hipSetDevice(g->device[s]) != hipSuccess ||
                        hipMemcpyAsync(d_q[root] + off, g->batches[s]->q, cnt * sizeof(double), hipMemcpyDeviceToDevice, g->batches[s]->stream) != hipSuccess ||
                        hipMemcpyAsync(d_qdot[root] + off, g->batches[s]->qd, cnt * sizeof(double), hipMemcpyDeviceToDevice, g->batches[s]->stream) != hipSuccess) {
                        (void)g_rccl.GroupEnd();
                        return fail(RMX_E_HIP, "rmx_group_gather_device: copying the root's own slice failed");
                    }
                    continue;
                }
                r = g_rccl.Send(g->batches[s]->q, cnt, ncclDouble, root, g->comms[s], g->batches[s]->stream);
                if (r == ncclSuccess) r = g_rccl.Send(g->batches[s]->qd, cnt, ncclDouble, root, g->comms[s], g->batches[s]->stream);
                if (r == ncclSuccess) r = g_rccl.Recv(d_q[root] + off, cnt, ncclDouble, s, g->comms[root], g->batches[root]->stream);
                if (r == ncclSuccess) r = g_rccl.Recv(d_qdot[root] + off, cnt, ncclDouble, s, g->comms[root], g->batches[root]->stream);
            }
        }
        const ncclResult_t re = g_rccl.GroupEnd();
        if (r != ncclSuccess) return gather_fail(g, r, g->last_gather);
        if (re != ncclSuccess) return gather_fail(g, re, "ncclGroupEnd");
    }
    for (int s = 0; s < S; ++s) {
        HIPCHK(hipSetDevice(g->device[s]));
        HIPCHK(hipStreamSynchronize(g->batches[s]->stream));
    }
    return RMX_OK;
}
extern "C" const char* rmx_group_gather_path(const rmx_group* g) { return g ? g->last_gather : ""; }
// the same gather into destinations the GROUP owns (for hosts that cannot allocate device memory themselves: the MEX gateway)
extern "C" int rmx_group_gather(rmx_group* g, int root_or_all) {
    if (!g) return fail(RMX_E_INVALID, "null group");
    const int S = (int)g->batches.size();
    if (root_or_all != RMX_GATHER_ALL && (root_or_all < 0 || root_or_all >= S)) return fail(RMX_E_INVALID, "rmx_group_gather: root must be a shard index or RMX_GATHER_ALL");
    if (g->gq.empty()) { g->gq.assign(S, nullptr); g->gqd.assign(S, nullptr); }
    const size_t bytes = (size_t)g->B * g->nr * sizeof(double);
    for (int s = 0; s < S; ++s) {
        if (!(root_or_all == RMX_GATHER_ALL || s == root_or_all)) continue;
        HIPCHK(hipSetDevice(g->device[s]));
        if (!g->gq[s]) HIPCHK(hipMalloc((void**)&g->gq[s], bytes ? bytes : 8));
        if (!g->gqd[s]) HIPCHK(hipMalloc((void**)&g->gqd[s], bytes ? bytes : 8));
    }
    return rmx_group_gather_device(g, g->gq.data(), g->gqd.data(), root_or_all);
}
extern "C" int rmx_group_gathered(rmx_group* g, int s, const double** d_q, const double** d_qdot) {
    if (!g || s < 0 || s >= (int)g->gq.size() || !g->gq[s] || !g->gqd[s]) return fail(RMX_E_INVALID, "rmx_group_gathered: shard holds no gathered state (rmx_group_gather first)");
    if (d_q) *d_q = g->gq[s];
    if (d_qdot) *d_qdot = g->gqd[s];
    return RMX_OK;
}
extern "C" int rmx_group_gathered_read(rmx_group* g, int s, double* q, double* qdot) {
    const double *dq = nullptr, *dqd = nullptr;
    if (int rc = rmx_group_gathered(g, s, &dq, &dqd)) return rc;
    const size_t bytes = (size_t)g->B * g->nr * sizeof(double);
    HIPCHK(hipSetDevice(g->device[s]));
    if (q) HIPCHK(hipMemcpy(q, dq, bytes, hipMemcpyDeviceToHost));
    if (qdot) HIPCHK(hipMemcpy(qdot, dqd, bytes, hipMemcpyDeviceToHost));
    return RMX_OK;
}

extern "C" int rmx_group_energy(rmx_group* g, double* T, double* V) {
    if (!g || !T || !V) return fail(RMX_E_INVALID, "null argument");
    for (size_t s = 0; s < g->batches.size(); ++s)
        if (int rc = rmx_energy(g->batches[s], T + g->first[s], V + g->first[s])) return rc;
    return RMX_OK;
}

// simLoop of the whole batch, first half: every shard's launch is in flight when this returns
extern "C" int rmx_group_step_async(rmx_group* g, const rmx_opts* opts, int nsteps, int integrator, int record) {
    if (!g) return fail(RMX_E_INVALID, "null group");
    if (integrator != 1 && integrator != 2) return fail(RMX_E_INVALID, "integrator must be 1 (BDF1) or 2 (BDF2)");
    for (rmx_batch* b : g->batches)
        if (int rc = rmx_stats_reset(b)) return rc;          // (enqueued on the shard's stream ahead of its launch)
    g->t0 = std::chrono::steady_clock::now();
    g->in_flight = true;
    for (rmx_batch* b : g->batches)
        if (int rc = rmx_step_history_async(b, opts, nsteps, integrator, record)) return rc;
    return RMX_OK;
}
// second half: wait for every shard, then the gather - each shard's slice of the counters and of the per-step record goes straight
// into its place in the caller's whole-batch arrays
extern "C" int rmx_group_sync(rmx_group* g, rmx_stats* stats, const rmx_history* hist) {
    if (!g) return fail(RMX_E_INVALID, "null group");
    int rc = RMX_OK;
    std::string first_err;
    for (rmx_batch* b : g->batches) {
        const int r = rmx_sync(b);          // wait for ALL of them even when one has failed: nothing may stay in flight
        if (r && !rc) { rc = r; first_err = g_err; }
    }
    if (g->in_flight) {
        g->wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - g->t0).count();
        g->in_flight = false;
    }
    if (rc) return fail(rc, first_err);
    for (size_t s = 0; s < g->batches.size(); ++s) {
        rmx_batch* b = g->batches[s];
        const size_t f = (size_t)g->first[s];
        if (hist && b->hist.nsteps > 0)
            if (int r = history_read(b, hist, (size_t)g->B, f)) return r;
        if (stats) {
            rmx_stats st{stats->newton_iters ? stats->newton_iters + f : nullptr, stats->ls_halvings ? stats->ls_halvings + f : nullptr,
                         stats->status ? stats->status + f : nullptr};
            if (int r = rmx_stats_read(b, &st)) return r;
        }
    }
    return RMX_OK;
}
extern "C" int rmx_group_step(rmx_group* g, const rmx_opts* opts, int nsteps, int integrator, rmx_stats* stats, const rmx_history* hist) {
    if (!g) return fail(RMX_E_INVALID, "null group");
    if (nsteps < 0) return fail(RMX_E_INVALID, "nsteps < 0");
    int record = 0;
    if (hist) {
        if ((hist->T == nullptr) != (hist->V == nullptr)) return fail(RMX_E_INVALID, "history T and V must be given together");
        if ((hist->q == nullptr) != (hist->qdot == nullptr)) return fail(RMX_E_INVALID, "history q and qdot must be given together");
        record = (hist->T ? RMX_REC_ENERGY : 0) | (hist->q ? RMX_REC_STATE : 0) | (hist->charts ? RMX_REC_CHARTS : 0);
    }
    const int rc = rmx_group_step_async(g, opts, nsteps, integrator, record);
    const int rs = rmx_group_sync(g, rc ? nullptr : stats, rc ? nullptr : hist);     // (also after a failed launch: drain what did start)
    return rc ? rc : rs;
}
extern "C" int rmx_group_timing(rmx_group* g, double* wall_ms, double* kernel_ms, double* start_ms, double* end_ms) {
    if (!g) return fail(RMX_E_INVALID, "null group");
    if (wall_ms) *wall_ms = g->wall_ms;
    for (size_t s = 0; s < g->batches.size(); ++s) {
        rmx_batch* b = g->batches[s];
        if (kernel_ms) kernel_ms[s] = b->last_ms;
        size_t ref = s;
        for (size_t r = 0; r < s; ++r)
            if (g->device[r] == g->device[s]) { ref = r; break; }
        float t0 = 0.f, t1 = (float)b->last_ms;
        if (ref != s) {
            HIPCHK(hipSetDevice(g->device[s]));
            if (hipEventElapsedTime(&t0, g->batches[ref]->ev0, b->ev0) != hipSuccess || hipEventElapsedTime(&t1, g->batches[ref]->ev0, b->ev1) != hipSuccess) {
                (void)hipGetLastError();
                t0 = 0.f; t1 = (float)b->last_ms;
            }
        }
        if (start_ms) start_ms[s] = t0;
        if (end_ms) end_ms[s] = t1;
    }
    return RMX_OK;
}
