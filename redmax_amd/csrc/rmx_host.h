// rmx_host.h -- what the host translation unit (redmax_hip.hip: the C ABI) and the kernel translation units
// (rmx_kernels.hip, compiled once per padded tree size RMX_NP in {4,8,16,32,64} so the builds run in parallel) share:
// launch argument blocks, the model / batch objects and the per-size launcher entry points.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "redmax_hip_profile.h"   /* redmax_hip.h + the measurement hooks */
#include "rmx_device.h"

using namespace rmx;

enum { INTEG_BDF1 = 1, INTEG_BDF2 = 2 };
constexpr size_t RMX_GARGS_BYTES = 2048;

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
    int* chart;       // [B][nsph] Euler charts of the spherical joints (in/out) or null
    double* histQ;    // [nsteps][B][nr] or null: q, qdot after every step (Scene.saveHistory, Scene.m:134-161)
    double* histQd;
    int* histC;       // [nsteps][B][nsph] or null: Euler charts after every step
    int* resume;      // [B] contact-capable kernels: first step the lean launch left to the launch with the contact terms
    unsigned long long* ticks;   // [B] or null: s_memtime ticks each rollout's wavefront spent in the launch(es) of this call (accumulated)
    // park and relaunch (rmx_device.h CoopCtx; null / 0: off): rollouts the launch with the contact terms gave up at the start of a step
    int* park;        // [1 + B + 3 B + 1]: [0] their number, [1 .. ] their indices + 1 in the order they parked (0: no entry yet),
                      // [1 + B + 3 traj ..] their pivot policy, [1 + 4 B] rollout workgroups that have finished (k_ground32)
    unsigned* xch;    // [ngroups][COOP_WORDS] exchange words of the cooperative groups (zero before the cooperative launch)
    int ngroups;      // cooperative groups in flight: group g finishes parked rollouts g, g + ngroups, ... one after the other
    int coop_map;     // measurement aid (RMX_COOP_MAP): 1 = member-major mapping of the cooperative launch's workgroups onto (group, member)
    int fused;        // the whole call in one launch (rmx_kernels.hip k_ground32): rollouts and cooperative groups side by side
    unsigned long long* xrec;   // [ngroups][2 COOP_REC] what the winner of a line search publishes to its group (rmx_ct32.h CoopPub; zero before the launch)
    int w2_noahead;   // two-wave tree kernels (rmx_kernels.hip w2_steps_bdf1): 1 = no evaluation is run ahead (RMX_W2_RUNAHEAD=0; tests)
    int pairc;        // the full 32-link chain under BDF1: 1 = the two-point kernel of rmx_pair32.h (default), 0 = the one-point kernel (RMX_PAIRC=0; tests)
};

struct AdjArgs {
    int B, nsteps, task_step, task_node;
    double xl[3], xt[3], pscale, wreg, wpos;
    double *q, *qd;
    double *qp, *qdp;         // BDF2: state of step k-1 after the rollout (out)
    const double* p;
    double *Hs, *Ms, *Ds;     // [B][nsteps][n*n]
    double* dPdq;             // [B][n]   dP/dq of the task step
    double* P;                // [B]
    double* dPdp;             // [B][nr]
    int *it, *status;
};

struct rmx_model {
    int device = 0;
    int n = 0, nr = 0, nm = 0, NP = 0;   // n: 1-DOF nodes on the device (after lowering multi-DOF joints)
    int nlist = 0;                      // joints/bodies in the caller's listing
    std::vector<int> idx_listing;   // reduced index per LISTED joint (-1 fixed)
    std::vector<int> node_of_listing;   // depth-first node index of each LISTED joint/body
    void* dbuf = nullptr;           // one device allocation holding all constant arrays
    void* dcon = nullptr;           // contact flags + cuboid sides (rmx_model_set_ground_contact)
    void* dsph = nullptr;           // axis variants of the spherical group nodes
    DevModel dm{};
    size_t smem_bytes = 0;
    int n_simd = 0;                 // SIMDs of the device (4 per CU)
    unsigned long long coop_ticks = 5000000000ull;   // ~2 s in s_memtime ticks of this device (DevOpts::coopTicks): measured at model creation
    int lds_limit = 0;              // LDS bytes one workgroup may hold (hipDeviceProp_t::sharedMemPerBlock); RMX_BIG_LDS_LIMIT overrides
    void* dgconst = nullptr;        // 64-lane plain models: the staged per-node constants in global memory (DevModel::gconst)
    int gconst_min_batch = 0;       // batches of at least this many rollouts run the global-constants kernels (0: never)
    int w2_max_batch = 0;           // 33..64-node trees / the full 32-link chain: batches of up to this many rollouts take two wavefronts each (0: never; RMX_W2_MAX)
    int adj_help_max_batch = 0;     // trees of <= 16 nodes: adjoint batches of up to this many rollouts take a second wavefront each for M, D (rmx_kernels.hip RMX_PART 8; 0: never)
    int w2_min_batch = 0;           // the full 32-link chain: ... and of at least this many (RMX_W2C_MIN; smaller batches keep the one-wave kernel every test pins)
    bool big = false;               // more than 64 nodes: the one-workgroup-per-tree kernels of rmx_big.hip
    bool pair32 = false;            // serial chain of <= 32 nodes with ForceGroundCuboid, no Euler-chart joints: the kernels around newton_pair
    std::vector<struct rmx_batch*> batches;   // live batches of this model (rmx_model_set_ground_contact drains their streams only)
};

struct rmx_batch {
    rmx_model* m = nullptr;
    int B = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double *q = nullptr, *qd = nullptr, *qp = nullptr, *qdp = nullptr;
    double *tmpA = nullptr, *tmpB = nullptr, *tmpC = nullptr;   // [B][nr] scratch for rmx_eval inputs
    int* started = nullptr;
    int* chart = nullptr;           // [B][nsph] current Euler chart of every spherical joint (JointSpherical.chart), 1..12
    int *it = nullptr, *ls = nullptr, *status = nullptr;
    int* resume = nullptr;          // [B] see StepArgs.resume
    int started_host = -1;          // what the device flag `started` holds (-1: not known): a step call rewrites it only when it changes
    int* park = nullptr;            // see StepArgs.park / xch (allocated for models whose steps can park: rmx_model::coop)
    unsigned* xch = nullptr;
    unsigned long long* xrec = nullptr;
    void* gargs = nullptr;          // RMX_GARGS_BYTES: the arguments of the fused ground launch (rmx_kernels.hip GroundArgs)
    int ngroups = 0;
    unsigned long long* ticks = nullptr;   // [B] see StepArgs.ticks (rmx_step_ticks)
    double* bigws = nullptr;        // trees of more than 64 nodes: per-rollout workspace of the rmx_big.hip kernels
    size_t bigws_stride = 0;        // doubles per rollout
    void* adjws = nullptr;          // rmx_adjoint_*: H, M, D of every step and rollout, dP/dq, P, dP/dp - one allocation that is kept
    size_t adjws_bytes = 0;         // between calls and only ever grows (hipMalloc + hipFree of 3 x 20 MB cost more than the kernels)
    double last_ms = 0.0;
    mutable const char* last_kernel = "";   // label of the step kernel the last step call launched (rmx_last_step_kernel; set by the launchers)
    bool async_pending = false;     // an rmx_step_*_async launch nobody has waited for yet (see pending_error_check)
    // per-step record of the last step call (Scene.saveHistory): device buffers, kept until the next step call so that an
    // asynchronous launch can be read back after rmx_sync (rmx_history_read)
    struct Hist {
        double *T = nullptr, *V = nullptr;      // [nsteps][B]
        double *Q = nullptr, *Qd = nullptr;     // [nsteps][B][nr]
        int* C = nullptr;                       // [nsteps][B][nsph]
        int nsteps = 0;
    } hist;                                     // the record of the last step call: null = that part was not recorded
    // The device buffers behind `hist` live on the batch and only grow: no hipFree (device-synchronising) sits between the launches
    // of shards that share a device, and a step call never pulls a buffer from under a launch still in flight.
    struct HistPool {
        double *T = nullptr, *V = nullptr, *Q = nullptr, *Qd = nullptr;
        int* C = nullptr;
        size_t capH = 0, capQ = 0, capC = 0;    // elements
    } hpool;
};

// launchers defined by rmx_kernels.hip for one RMX_NP each
#define RMX_CAT_(a, b) a##b
#define RMX_CAT(a, b) RMX_CAT_(a, b)
#define RMX_DECLARE_LAUNCHERS(NPV) \
    void RMX_CAT(launch_eval_, NPV)(const rmx_model* m, const rmx_batch* b, bool wantH, double eta, double* dg, double* dH); \
    void RMX_CAT(launch_step_np_, NPV)(const rmx_model* m, const rmx_batch* b, int integ, const DevOpts& o, const StepArgs& a); \
    void RMX_CAT(launch_euler_, NPV)(const rmx_model* m, const rmx_batch* b, double h, const StepArgs& a); \
    void RMX_CAT(launch_energy_, NPV)(const rmx_model* m, const rmx_batch* b, double* dT, double* dV); \
    void RMX_CAT(launch_adjoint_, NPV)(const rmx_model* m, const rmx_batch* b, int integ, const DevOpts& o, const AdjArgs& a); \
    void RMX_CAT(launch_phase_, NPV)(const rmx_model* m, const rmx_batch* b, int reps, double h, unsigned long long* d); \
    void RMX_CAT(launch_mfd_, NPV)(const rmx_model* m, const rmx_batch* b, double* dM, double* df, double* dD); \
    void RMX_CAT(launch_mfd_ct_, NPV)(const rmx_model* m, const rmx_batch* b, double* dM, double* df, double* dD); \
    void RMX_CAT(launch_eval_ct_, NPV)(const rmx_model* m, const rmx_batch* b, bool wantH, double eta, double* dg, double* dH); \
    void RMX_CAT(launch_step_ct_, NPV)(const rmx_model* m, const rmx_batch* b, int integ, const DevOpts& o, const StepArgs& a); \
    void RMX_CAT(launch_energy_ct_, NPV)(const rmx_model* m, const rmx_batch* b, double* dT, double* dV); \
    void RMX_CAT(launch_step_fullchain_, NPV)(const rmx_model* m, const rmx_batch* b, int integ, const DevOpts& o, const StepArgs& a);
// 64-lane plain step kernels reading the per-node constants from global memory (rmx_kernels.hip RMX_PART 3) and the staging kernel
void launch_step_gconst_64(const rmx_model* m, const rmx_batch* b, int integ, const DevOpts& o, const StepArgs& a);
void launch_step_fulln_64(const rmx_model* m, const rmx_batch* b, int integ, const DevOpts& o, const StepArgs& a);
void launch_step_w2_64(const rmx_model* m, const rmx_batch* b, int integ, const DevOpts& o, const StepArgs& a);
void launch_step_w2c_32(const rmx_model* m, const rmx_batch* b, const DevOpts& o, const StepArgs& a);
// rmx_kernels.hip RMX_PART 7: the full 32-link chain, BDF1, two points per evaluation (rmx_pair32.h)
void launch_step_pairchain_32(const rmx_model* m, const rmx_batch* b, const DevOpts& o, const StepArgs& a);
void launch_adjoint_help_16(const rmx_model* m, const rmx_batch* b, int integ, const DevOpts& o, const AdjArgs& a);
void launch_phase_pairchain_32(const rmx_model* m, const rmx_batch* b, int reps, double h, unsigned long long* d);
// rmx_kernels.hip RMX_PART 4 (32 lanes): serial chains with ground contact - the launch with the contact terms around newton_pair
// (rmx_ct32.h) and the cooperative launch that finishes the rollouts it parked
void launch_step_pair_32(const rmx_model* m, const rmx_batch* b, int integ, const DevOpts& o, const StepArgs& a, bool fused);
void launch_stage_consts_64(const rmx_model* m, double* dst, hipStream_t stream);
// rmx_big.hip: trees of 65..BIG_MAXN nodes, one workgroup per rollout
size_t big_ws_doubles(const rmx_model* m);
void launch_big_step(const rmx_model* m, const rmx_batch* b, int integ, const DevOpts& o, const StepArgs& a);
void launch_big_eval(const rmx_model* m, const rmx_batch* b, bool wantH, double eta, double* dg, double* dH);
void launch_big_energy(const rmx_model* m, const rmx_batch* b, double* dT, double* dV);
RMX_DECLARE_LAUNCHERS(4)
RMX_DECLARE_LAUNCHERS(8)
RMX_DECLARE_LAUNCHERS(16)
RMX_DECLARE_LAUNCHERS(32)
RMX_DECLARE_LAUNCHERS(64)
