// rmx_device.h -- device-side code of the batched RedMax implicit step for gfx950 (MI355X).
//
// One 64-lane wavefront owns one trajectory.  Within the wave, lane j owns joint/body j of the
// kinematic tree (depth-first order), so every per-node quantity lives in registers and the tree
// recursions of the reference (Joint.update / Joint.computeJacobian / Body.computeMassGrav,
// matlab-diff/+redmax/Joint.m:382-613, Body.m:70-135) become
//   * root->node path products/sums   : serial chains: DPP row scans + row hand-over; trees: pointer jumping (ds_bpermute)
//   * node->leaves subtree sums       : transpose through LDS, two lanes per component scan the nodes
//   * the nr x nr Hessian             : two products on the fp64 matrix cores (v_mfma_f64_16x16x4_f64), masked by the ancestor /
//                                       descendant relation (n <= 32: 32x32; 33..64 nodes, plain evaluation: 64x64 in two column
//                                       halves); padded sizes < 32 and the contact terms of 33..64 nodes: lane = row, v_readlane columns
//   * the dense solve  dx = -H\g      : diagonal pivots under a growth guard, every update one DPP-fused FMA (n <= 32: column-split
//                                       first half + row-per-lane tail; 33..64 rows: block columns of 16, H resident in LDS);
//                                       partial pivoting on demand: lane = row, pivot row by batched v_readlane broadcasts
// One wave per SIMD means the kernel time is the instruction count on the path: see DESIGN.md "The instruction-count pass".
// The algebra (world-frame recursive Newton-Euler with analytic derivatives, no J / dJdq tensors)
// is derived in DESIGN.md and restated executable in tests/proto_worldframe.py.
#pragma once
#include <hip/hip_runtime.h>

// Synchronisation and LDS layout hooks.  One 64-lane wavefront per trajectory; the defaults are a workgroup barrier (which the
// compiler reduces to the LDS wait for a one-wave workgroup) and the per-node constants right behind the wave's scratch.  A
// translation unit whose workgroups hold several independent wavefronts redefines both before including this header: wave-local
// ordering only, and one copy of the constants shared by the waves of the workgroup.
#ifndef RMX_SYNC
#define RMX_SYNC() __syncthreads()
#endif
// RMX_W2 (a translation unit compiled with it, rmx_kernels.hip RMX_PART 5): workgroups of TWO wavefronts per 64-node tree.  Wave 0 is
// the rollout; wave 1 joins it for the Hessian tiles (its column half) and for the block-column elimination (its share of the later
// column blocks of every phase) and waits at a workgroup barrier otherwise.  RMX_SYNC() is wave-local ordering in that unit (the
// front, the pivoting fallback and everything else belong to wave 0 alone); RMX_WG_BAR() is the barrier both waves meet at.
#ifndef RMX_W2
#define RMX_W2 0
#endif
#define RMX_WG_BAR() __syncthreads()
#ifndef RMX_W2_FULL_HELPER
#define RMX_W2_FULL_HELPER 0      // 1: the helper wave stays in the solve to its end (back substitution included, result dropped)
#endif
#if RMX_W2
// (the base of the dynamic LDS: RMX_CONSTS of that translation unit must not depend on which scratch an evaluation works on)
__device__ __forceinline__ double* rmx_smem_base() {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    return smem;
}
// __syncthreads() without the s_barrier: LDS traffic of ONE wavefront is ordered by the wait alone
__device__ __forceinline__ void rmx_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
#endif
// RMX_SYNC() for a translation unit whose workgroups are ONE wavefront (-DRMX_SYNC()=rmx_lane_sync()): the LDS executes the DS
// instructions of a wavefront in issue order, so a read that follows a write in program order sees it whatever lanes are involved -
// what is needed is that the COMPILER keeps the order (a wavefront-scope fence: no code), not that the wavefront waits for its own
// writes to be acknowledged (what __syncthreads() / a workgroup-scope fence compile to: s_waitcnt lgkmcnt(0), a full LDS round trip
// at every hand-over; the waits a read's RESULT needs are the register dependencies the compiler tracks anyway).
__device__ __forceinline__ void rmx_lane_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#ifndef RMX_LANE_SYNC_NO_WB
    __builtin_amdgcn_wave_barrier();
#endif
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// RMX_GLOBAL_CONSTS (a translation unit compiled with it, rmx_kernels.hip RMX_PART 3): the per-node constants are NOT staged in LDS
// but read from a table in global memory (DevModel::gconst, same [row][node] layout, L2-resident and shared by the whole batch).
// A 64-lane tree needs 33.8 KB of scratch (H for the block-column solve) plus 34.8 KB of constants per wavefront: two wavefronts
// per CU; without the constants four fit, and a batch of more than two rollouts per CU runs in half the time.
#ifndef RMX_CONSTS
#ifdef RMX_GLOBAL_CONSTS
#define RMX_CONSTS(sAcc, n, NP) (M.gconst)
#else
#define RMX_CONSTS(sAcc, n, NP) ((sAcc) + acc_doubles((n), (NP)))
#endif
#endif

namespace rmx {

constexpr int MAXN = 64;          // nodes per tree handled by one wavefront
constexpr int BIG_MAXN = 256;     // nodes per tree handled by one workgroup (rmx_big.hip: the general kernels for trees of more than 64 nodes)
constexpr int BIG_MAXROUNDS = 8;  // log2(BIG_MAXN)
constexpr int NACC = 28;          // subtree-accumulated numbers per body (w6, m1, mc3, Ibar6, TL9, hf3)
constexpr int ACC_STRIDE = 29;    // odd stride: conflict-free lane=node LDS writes
constexpr int NCOL = 18;          // column-side Hessian vectors per node (yz6, m1 6, m2w3, sw3)
constexpr int COL_STRIDE = 19;
constexpr int MAXROUNDS = 6;      // log2(MAXN)
constexpr int NCON = 14;          // rows of DevModel::con
constexpr int NGROUND = 10;       // of which per-body ground frame and constants (GroundC)
constexpr int NCONST = 68;        // per-node constants staged in LDS: K(36) sb(6) I4(4) prm(8) type(1) rel(2) anc(6) end(1) contact(1) sides(3)
constexpr int MAXSPH = 85;        // spherical joints per tree (three nodes each): BIG_MAXN / 3; the one-wavefront kernels (<= 64 nodes) see at most 21
constexpr int SPH_ROWS = 42;      // per-node constants that depend on a spherical node's axis: K(36) + sb(6), the first LDS rows
// doubles of the accumulation scratch at the start of a wavefront's LDS: (n+1) rows of the subtree scan, or the Hessian's
// column vectors [NP][NCOLX], whichever is larger; the per-node constants follow it
constexpr int HM_OP_STRIDE = 33;  // MFMA Hessian (NP = 32): operand rows [k][node], odd stride in doubles
constexpr int HM_ROWS = 57;       // RU 8, RL 12|20, CU 8, CL 12|20, Hdiag 1 (|: with ground contact)
constexpr int HM_H_STRIDE = 34;   // H staged row-major [32][34] for the hand-over to row-per-lane (16-byte aligned rows)
// trees of 33..64 nodes (NP = 64): operand rows RU 6, RL 12, CU 6, CL 12, Hdiag 1 of [k][node] with stride 65, then H row-major
// [64][66] (H64_STRIDE; column 64: the right-hand side); the front's scratch shares the area
constexpr int H64_OP_STRIDE = 65, H64_OP_ROWS = 37;
__host__ __device__ constexpr int acc_doubles(const int n, const int NP) {
    const int a = (n + 1) * ACC_STRIDE;
    const int b = NP == 32 ? HM_ROWS * HM_OP_STRIDE : 0;     // 1881 doubles; also covers H: 32*34 = 1088
    const int c = NP == 64 ? 64 * 66 : 0;                    // lu_solve_neg_diag64's staging of H (H64_STRIDE)
    return (a > b ? a : b) > c ? (a > b ? a : b) : c;
}
constexpr bool HESS_MFMA = true;  // n <= 32: Hessian assembly on the fp64 matrix cores (false: half-wave split of the column loop)
constexpr bool LU_DPP_TAIL = true;                      // guarded LU: last 16 pivots with the broadcast fused into the FMA (DPP)
constexpr bool LU_SPLIT32 = HESS_MFMA && LU_DPP_TAIL;   // n <= 32: pivots 0..15 in the column-split layout of lu_solve_neg_diag32
constexpr bool LU_SPLIT64 = LU_DPP_TAIL;                // 33..64 rows: the block-column layout of lu_solve_neg_diag64
constexpr bool HESS_MFMA64 = LU_SPLIT64;                // 33..64 nodes, plain models, one wavefront: Hessian on the matrix cores, H left in LDS
constexpr int H64_STRIDE = 66;    // lu_solve_neg_diag64: H staged row-major [64][66] (column 64: right-hand side; 16-byte aligned rows)
// Column stride of the per-node constants in LDS.  Trees padded to fewer than 64 lanes get one extra "idle" column (index NP):
// identity joint transform, zero everything else.  Lanes beyond the padded size read it, idle node slots n..NP-1 hold the same
// defaults in their own columns, so the evaluation loads constants without any per-lane selects.
__host__ __device__ constexpr int cstride(const int NP) { return NP < 64 ? NP + 1 : NP; }
constexpr int NCOLX = 24;         // with ground contact the column side also needs m2v(3) and sv(3)

// Constant per-model data, SoA over nodes (stride MAXN) so lane=node loads coalesce.
struct DevModel {
    int n;            // nodes (joints == bodies)
    int stride;       // node stride of the SoA arrays below: MAXN, or BIG_MAXN for trees of more than 64 nodes
    int nr;           // reduced DOFs
    int rounds;       // pointer-jumping rounds = ceil(log2(max depth + 1))
    int is_chain;     // every subtree ends at n (serial chain): skip the range subtraction
    const double* K;  // [36][MAXN]  T_j(q) = K0 + u K1 + w K2, rows: R(9) then p(3) for K0,K1,K2
    const double* sb; // [6][MAXN]   joint screw in the body frame, A0_ij * S  (Joint.m:508)
    const double* I4; // [4][MAXN]   I1,I2,I3,m  (se3.inertiaCuboid)
    const double* prm;// [8][MAXN]   tau, stiffness, damping, qRest, qLimL, qLimU, qLimK, qLimD
    const int* type;  // [MAXN]
    const int* idx;   // [MAXN] reduced index (reference leaf-to-root numbering) or -1
    const int* end;   // [MAXN] one past the last node of the subtree (depth-first order)
    const int* anc;   // [MAXROUNDS][MAXN] ancestor 2^r levels up, or -1
    const unsigned long long* rel;  // [2][MAXN] bit i of rel[j]: node i is a strict ancestor of j; of rel[MAXN+j]: strict descendant
    double grav[3];
    // ForceGroundCuboid: every flagged body carries its own force object (null con: no contact forces in the scene)
    const double* con;   // [NCON][MAXN]  contact flag, cuboid sides(3) | per body: plane normal (Z axis of ITS ground frame) (3) and origin
                         // (3) ForceGroundCuboid.m:56-57, kn, kt (setStiffness), mu (setFriction), kd (setDamping).  Rows 0..3 travel with
                         // the per-node constants (NCONST), rows 4..13 are staged behind them by con_setup in the contact-capable kernels
    // JointSpherical / JointFree3D: group g = nodes sph_first[g] .. +2 (revolute about the axes of the group's Euler chart)
    const double* gconst;   // [NCONST][cstride(NP)] the per-node constants as smem_setup stages them, in global memory (RMX_GLOBAL_CONSTS)
    int nsph;
    const double* sphV;  // [nsph][3 nodes][3 axes][SPH_ROWS]  K and sb of each group node for axis x, y, z
    short sph_first[MAXSPH + 3];
    // Branching trees of 33..64 nodes and depth <= TREE_DMAX: the structure the multifrontal solve walks (tree_solve64).  tree: [TREE_ROWS][MAXN]
    // ints - row 0 depth (-1: no node), rows 1 .. TREE_DMAX the ancestor at depth 0 .. TREE_DMAX - 1 (or -1), then the children, then
    // the parent; tree_dmax = 0: the dense solve (serial chains, deep trees, other sizes)
    const int* tree;
    int tree_dmax, tree_cmax;
};
constexpr int TREE_DMAX = 7;       // deepest node of a tree the multifrontal solve takes (levels 0 .. 7: frontal matrices of up to 8 x 9)
constexpr int TREE_CMAX = 4;       // most children of a node
constexpr int TREE_ROWS = 1 + TREE_DMAX + TREE_CMAX + 1;

struct DevOpts {
    double h, tol, dxMax;
    int iterMax, iterLsMax;
    int lu_mode;      // 0: diagonal pivots under a growth guard, partial pivoting on demand; 1: always partial pivoting
    double comp;      // 1.0: compensated Newton iterate (x + xlo, see newton_impl); 0.0: plain doubles (the reference's lattice)
    int lsFailLimit;  // rmx_opts.ls_fail_limit: > 0 ends a step's Newton loop at its N-th failed line search
    unsigned long long coopTicks;   // how long a member of a cooperative group waits for its group before it gives the rollout up: ~2 s in
                      // s_memtime ticks of THIS device (rmx_model::coop_ticks - the counter runs at the shader clock on gfx950, at a
                      // constant 100 MHz on older parts: the host derives the budget, the kernels do not assume a rate)
    int parkHalv;     // > 0 (contact kernels of serial chains of <= 32 nodes): a solve whose line searches have spent more than this many
                      // halvings parks its rollout at the start of the step for the cooperative launch (newton_impl COOP); 0: never
};

// ----------------------------------------------------------------------------- small helpers

__device__ __forceinline__ void cross3(const double a[3], const double b[3], double c[3]) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ double dot3(const double a[3], const double b[3]) {
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}
// y = R x, R row-major 3x3
__device__ __forceinline__ void mat3v(const double R[9], const double x[3], double y[3]) {
    y[0] = R[0] * x[0] + R[1] * x[1] + R[2] * x[2];
    y[1] = R[3] * x[0] + R[4] * x[1] + R[5] * x[2];
    y[2] = R[6] * x[0] + R[7] * x[1] + R[8] * x[2];
}
// symmetric 3x3 stored xx,xy,xz,yy,yz,zz
__device__ __forceinline__ void sym3v(const double S[6], const double x[3], double y[3]) {
    y[0] = S[0] * x[0] + S[1] * x[1] + S[2] * x[2];
    y[1] = S[1] * x[0] + S[3] * x[1] + S[4] * x[2];
    y[2] = S[2] * x[0] + S[4] * x[1] + S[5] * x[2];
}

__device__ __forceinline__ double shfl_d(double v, int src) { return __shfl(v, src, 64); }

// v_writelane_b32: lane l of the result takes the scalar s, the other lanes keep old.  (This clang has no __builtin_amdgcn_writelane;
// the intrinsic is reached by its IR name, so the compiler still sees an ordinary instruction and adds the wait states after a
// v_readlane that produced s.)
extern "C" __device__ int rmx_llvm_writelane_i32(int, int, int) __asm("llvm.amdgcn.writelane.i32");
__device__ __forceinline__ int writelane_i(const int s, const int l, const int old) { return rmx_llvm_writelane_i32(s, l, old); }

__device__ __forceinline__ double readlane_d(double v, int l) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, l);
    hi = __builtin_amdgcn_readlane(hi, l);
    return __hiloint2double(hi, lo);
}

// acc -= m * (value of src in lane N of the caller's own 16-lane row): v_fmac_f64 with the DPP row_newbcast control, the one
// DPP control gfx90a+ allows on 64-bit VALU operations.  The broadcast rides on the FMA: 7.3 cycles per element for a lone
// wavefront against 15.4 for two v_readlane plus an FMA (tools/ubench.hip modes 14, 16, 17).  The compiler emits
// v_mov_b64_dpp + v_fmac for the builtin form (13.1 cycles) and does not fold them, hence the assembly.
// Hazard: a DPP read of a VGPR written by one of the two preceding VALU instructions returns the OLD value, and the
// assembler adds no wait states.  Every use below reads lane N of a register that the interfering write leaves unchanged IN
// LANE N (the pivot lane's multiplier is 0, so its rows are rewritten with the same values), so old and new agree.
// Where a broadcast does read a value that the previous step changed (the look-ahead column and the right-hand side in
// lu_diag_tail), a whole step's instructions lie in between, except in the last two steps, which take NOPS (s_nop 1 first).
template <int N, bool NOPS = false>
__device__ __forceinline__ void fmsub_rowbcast(double& acc, const double src, const double m) {
    static_assert(N >= 0 && N < 16, "row_newbcast lane");
    if constexpr (NOPS)
        asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(m), "n"(N));
    else
        asm volatile("v_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(m), "n"(N));
}

// acc += m * (value of src in lane N of the caller's own 16-lane row).  Same instruction and hazard note as fmsub_rowbcast; the
// column loops that use it (eval_hess / eval_MD up to 16 nodes) read registers written long before the loop.
template <int N>
__device__ __forceinline__ void fmadd_rowbcast(double& acc, const double src, const double m) {
    static_assert(N >= 0 && N < 16, "row_newbcast lane");
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(m), "n"(N));
}

// The DPP hazard above, for operands the compiler is free to compute late: every v[c] is materialised (an empty volatile asm takes it
// as an operand; volatile asm statements keep their order) and two wait states pass before the first broadcast that follows.
template <int N>
__device__ __forceinline__ void dpp_settle(double (&v)[N]) {
#pragma unroll
    for (int c = 0; c < N; ++c) asm volatile("" : "+v"(v[c]));
    asm volatile("s_nop 1");
}

// the value of lane N of the caller's own 16-lane row, in every lane (v_mov_b64_dpp row_newbcast; the compiler takes care of the
// wait states after a VALU write of the source)
template <int N>
__device__ __forceinline__ double row_bcast(const double v) {
    static_assert(N >= 0 && N < 16, "row_newbcast lane");
    return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + N, 0xF, 0xF, true);
}

// v_permlane32_swap (gfx950): swaps lanes 32..63 of its first operand with lanes 0..31 of the second.
// dup_lo: every lane l >= 32 receives the value of lane l-32 (lanes < 32 keep theirs); take_hi: every lane l < 32 receives
// the value of lane l+32.
__device__ __forceinline__ double dup_lo(const double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double(b[0], a[0]);
}
__device__ __forceinline__ double take_hi(const double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double(b[1], a[1]);
}

// DPP lane permutations inside a 16-lane row (no LDS traffic, a few cycles each)
constexpr int DPP_XOR1 = 0xB1;         // quad_perm [1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;         // quad_perm [2,3,0,1]
constexpr int DPP_HALF_MIRROR = 0x141; // lane i <-> 7-i within each 8 lanes
constexpr int DPP_MIRROR = 0x140;      // lane i <-> 15-i within the row
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
    return __hiloint2double(dpp_i<CTRL>(__double2hiint(v)), dpp_i<CTRL>(__double2loint(v)));
}

// Sum over the 64 lanes, identical (bitwise) in every lane: butterfly inside each 16-lane row with DPP, then the
// four row sums are combined through scalar registers in a fixed order.
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_d<DPP_XOR1>(v);
    v += dpp_d<DPP_XOR2>(v);
    v += dpp_d<DPP_HALF_MIRROR>(v);
    v += dpp_d<DPP_MIRROR>(v);
    return (readlane_d(v, 0) + readlane_d(v, 16)) + (readlane_d(v, 32) + readlane_d(v, 48));
}
// the same for a vector that is zero on the lanes past the padded tree size NP (residuals, Newton updates: idle lanes carry exact
// zeros): only the 16-lane rows that can hold something are read out.  x + 0 = x, so the sum has the bits of wave_sum.
template <int NP>
__device__ __forceinline__ double wave_sum_np(double v) {
    if constexpr (NP > 32) {
        return wave_sum(v);
    } else {
        v += dpp_d<DPP_XOR1>(v);
        v += dpp_d<DPP_XOR2>(v);
        v += dpp_d<DPP_HALF_MIRROR>(v);
        v += dpp_d<DPP_MIRROR>(v);
        if constexpr (NP > 16) return readlane_d(v, 0) + readlane_d(v, 16);
        else return readlane_d(v, 0);
    }
}
__device__ __forceinline__ unsigned wave_umax(unsigned v) {
    unsigned t;
    t = (unsigned)dpp_i<DPP_XOR1>((int)v); v = t > v ? t : v;
    t = (unsigned)dpp_i<DPP_XOR2>((int)v); v = t > v ? t : v;
    t = (unsigned)dpp_i<DPP_HALF_MIRROR>((int)v); v = t > v ? t : v;
    t = (unsigned)dpp_i<DPP_MIRROR>((int)v); v = t > v ? t : v;
    const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
    const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
    const unsigned ab = a > b ? a : b, cd = c > d ? c : d;
    return ab > cd ? ab : cd;
}
// same, when only lanes 0..31 can hold candidates (NP <= 32): two rows
__device__ __forceinline__ unsigned wave_umax32(unsigned v) {
    unsigned t;
    t = (unsigned)dpp_i<DPP_XOR1>((int)v); v = t > v ? t : v;
    t = (unsigned)dpp_i<DPP_XOR2>((int)v); v = t > v ? t : v;
    t = (unsigned)dpp_i<DPP_HALF_MIRROR>((int)v); v = t > v ? t : v;
    t = (unsigned)dpp_i<DPP_MIRROR>((int)v); v = t > v ? t : v;
    const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
    return a > b ? a : b;
}
// row_shr:K inside each 16-lane row; lanes without a source (lane&15 < K) receive 0
template <int K>
__device__ __forceinline__ double dpp_shr0(double v) {
    // bound_ctrl: lanes without a source read 0 from the DPP itself; with bound_ctrl off the destination would have to be
    // pre-loaded with the fill value (two v_mov + a hazard s_nop per use)
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x110 + K, 0xF, 0xF, true);
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x110 + K, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
// row_shr:K with a caller-chosen value for the lanes without a source
template <int K>
__device__ __forceinline__ double dpp_shr_old(double v, const double old) {
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), 0x110 + K, 0xF, 0xF, false);
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), 0x110 + K, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
// row_shl:K inside each 16-lane row (lane l receives lane l+K); lanes without a source receive 0
template <int K>
__device__ __forceinline__ double dpp_shl0(double v) {
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x100 + K, 0xF, 0xF, true);
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x100 + K, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
// 1/x to full double precision (LAPACK's dgetf2 also scales the pivot column by the reciprocal): the hardware estimate r
// is good to 2^-24.4, one cubic step r (1 + e + e^2), e = 1 - x r, leaves e^3 = 2^-73 before rounding: three FMAs on the
// pivot's critical path instead of the four of two Newton steps, and the same bits (tools/rcp_accuracy.hip: both equal the
// correctly rounded 1/x on 2^20 samples over 60 binades)
__device__ __forceinline__ double recip(double x) {
    const double r = __builtin_amdgcn_rcp(x);
    const double e = fma(-x, r, 1.0);
    return fma(r, fma(e, e, e), r);
}

// an = sqrt(a2) and ian = 1 / sqrt(a2) from ONE v_rsq_f64 (good to ~2^-24): a cubic step r (1 + u/2 + 3 u^2/8), u = 1 - a2 r^2,
// leaves u^3 ~ 2^-72 before rounding; sqrt = a2 r with one residual correction.  10 instructions for both against the 22 + 14 of
// the library sqrt (range scaling, class test) and the IEEE division - per penetrating corner of the ground contact, on the path
// of every line-search trial.  a2 = 0: an = 0, ian = NaN (the caller's 0 * ian is NaN like its 0 * (1/0) was).
__device__ __forceinline__ void sqrt_rcp(const double a2, double& an, double& ian) {
    const double r0 = __builtin_amdgcn_rsq(a2);
    const double t = a2 * r0;
    const double u = fma(-t, r0, 1.0);
    const double r = fma(r0, u * fma(0.375, u, 0.5), r0);
    const double s = a2 * r;
    const double e = fma(-s, s, a2);
    const double sq = fma(0.5 * e, r, s);
    an = a2 > 0.0 ? sq : 0.0;
    ian = r;
}

// s + e = a + b exactly (Knuth's TwoSum, no ordering assumption; six additions that must not be re-associated)
__device__ __forceinline__ void two_sum(const double a, const double b, double& s, double& e) {
    s = a + b;
    const double bb = s - a;
    e = (a - (s - bb)) + (b - bb);
}

// Per-lane (per-node) results of one evaluation that the caller keeps.
struct NodeOut {
    double g;        // residual entry of this node's DOF (0 for a fixed joint)
    double eT, eV;   // kinetic / potential energy contribution of this body + joint
};

// ----------------------------------------------------------------------------- the evaluation
//
// evalBDF1 / computeValues (driverRedMaxBDF1.m:160-243) for the generic implicit residual
//     qdot = (x - qA)/eta ; v = x - qB ; g = M v - eta^2 f ; H = dg/dx
// split in two stages that share the per-node state in registers:
//   eval_front : kinematics, path sums, body wrenches, subtree sums, residual g   (the reference's nargout==1 path)
//   eval_hess  : the Hessian row of this node from the retained state              (the extra work of nargout==2)
// The reference evaluates g at the accepted line-search point and then g AND H again at the same point at the top of the
// next Newton iteration (driverRedMaxBDF1.m:103,125); here the second evaluation reuses the first one's state.
#define RMX_STAMP(k)                                                  \
    if (TIMED) {                                                      \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        stamps[k] += now_ - last_;                                    \
        last_ = now_;                                                 \
    }

// What eval_hess needs from eval_front (all per lane = per node, world frame)
struct FrontState {
    double sw[3], sv[3];      // joint screw
    double phw[3], phv[3];    // phi  = (J qdot)_j
    double xiw[3], xiv[3];    // xi   = ad(phi) s
    double bw[3], bv[3];      // beta = (J v + eta^2 Jdot qdot)_j
    double S[NACC];           // subtree sums: W(6), m, mc(3), Ibar(6), TL(9), hf(3)
    double eta, kd, dd;       // step; -Kr and -Dr of this joint (stiffness/damping incl. active limits)
    double Rw[9], pw[3];      // world transform of this body (only kept alive where a caller reads it)
    bool touched;             // ground contact: some corner of some body of this tree penetrates (wave-uniform)
    double tau_add = 0.0;     // extra joint torque set by the caller (adjoint task parameters, TaskBDF1PointPos.applyStep)
    unsigned long long anc_m, desc_m;   // bit i: node i is a strict ancestor / descendant of this node
    bool act, dof;
};

// Inclusive root->node "path" composition for a serial chain (parent(j) = j-1): Hillis-Steele scan with DPP row shifts
// inside each 16-lane row, then the row prefixes are chained through scalar broadcasts.  NROWS = rows in use.
template <int NP>
__device__ __forceinline__ void chain_scan_sum6(const int lane, double (&a)[3], double (&b)[3]) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        a[c] += dpp_shr0<1>(a[c]); b[c] += dpp_shr0<1>(b[c]);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        a[c] += dpp_shr0<2>(a[c]); b[c] += dpp_shr0<2>(b[c]);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        a[c] += dpp_shr0<4>(a[c]); b[c] += dpp_shr0<4>(b[c]);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        a[c] += dpp_shr0<8>(a[c]); b[c] += dpp_shr0<8>(b[c]);
    }
    constexpr int NROWS = (NP + 15) / 16;
#pragma unroll
    for (int r = 1; r < NROWS; ++r) {   // row r adds the (already complete) total of lane 16r-1
        // a 0/1 weight per lane and one FMA with the scalar broadcast as operand (exact: 1*t + a, 0*t + a) instead of moving
        // the broadcast into a vector register and selecting; all six broadcasts first, then the FMAs (readlane -> use hazard)
        const double w = ((lane >> 4) == r) ? 1.0 : 0.0;
        double ta[3], tb[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            ta[c] = readlane_d(a[c], 16 * r - 1);
            tb[c] = readlane_d(b[c], 16 * r - 1);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            a[c] = fma(w, ta[c], a[c]);
            b[c] = fma(w, tb[c], b[c]);
        }
    }
}

template <int K>
__device__ __forceinline__ void chain_compose_step(const int lane, double (&R)[9], double (&p)[3]) {
    // lanes without a predecessor K lanes down the row receive the identity transform, so the composition below needs no
    // lane condition (1*x + 0*y + 0*z reproduces x exactly)
    (void)lane;
    double Ra[9], pa[3];
#pragma unroll
    for (int c = 0; c < 9; ++c) Ra[c] = (c % 4 == 0) ? dpp_shr_old<K>(R[c], 1.0) : dpp_shr0<K>(R[c]);
#pragma unroll
    for (int c = 0; c < 3; ++c) pa[c] = dpp_shr0<K>(p[c]);
    double Rn[9], pn[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int k = 0; k < 3; ++k) Rn[3 * i + k] = Ra[3 * i] * R[k] + Ra[3 * i + 1] * R[3 + k] + Ra[3 * i + 2] * R[6 + k];
        pn[i] = Ra[3 * i] * p[0] + Ra[3 * i + 1] * p[1] + Ra[3 * i + 2] * p[2] + pa[i];
    }
#pragma unroll
    for (int c = 0; c < 9; ++c) R[c] = Rn[c];
#pragma unroll
    for (int c = 0; c < 3; ++c) p[c] = pn[c];
}
template <int NP>
__device__ __forceinline__ void chain_scan_transform(const int lane, double (&R)[9], double (&p)[3]) {
    chain_compose_step<1>(lane, R, p);
    chain_compose_step<2>(lane, R, p);
    chain_compose_step<4>(lane, R, p);
    chain_compose_step<8>(lane, R, p);
    constexpr int NROWS = (NP + 15) / 16;
#pragma unroll
    for (int r = 1; r < NROWS; ++r) {
        double Ra[9], pa[3];
#pragma unroll
        for (int c = 0; c < 9; ++c) Ra[c] = readlane_d(R[c], 16 * r - 1);
#pragma unroll
        for (int c = 0; c < 3; ++c) pa[c] = readlane_d(p[c], 16 * r - 1);
        if ((lane >> 4) == r) {
            double Rn[9], pn[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
#pragma unroll
                for (int k = 0; k < 3; ++k) Rn[3 * i + k] = Ra[3 * i] * R[k] + Ra[3 * i + 1] * R[3 + k] + Ra[3 * i + 2] * R[6 + k];
                pn[i] = Ra[3 * i] * p[0] + Ra[3 * i + 1] * p[1] + Ra[3 * i + 2] * p[2] + pa[i];
            }
#pragma unroll
            for (int c = 0; c < 9; ++c) R[c] = Rn[c];
#pragma unroll
            for (int c = 0; c < 3; ++c) p[c] = pn[c];
        }
    }
}

// Subtree sums for a serial chain = suffix sums over the lanes: DPP row_shl scan inside each row, then the complete total of
// the next row's first lane is handed down through scalar registers (highest row first).  No LDS, no barriers.
template <int NP, int NS>
__device__ __forceinline__ void chain_suffix_sum(const int lane, double (&S)[NACC]) {
#pragma unroll
    for (int c = 0; c < NS; ++c) S[c] += dpp_shl0<1>(S[c]);
#pragma unroll
    for (int c = 0; c < NS; ++c) S[c] += dpp_shl0<2>(S[c]);
#pragma unroll
    for (int c = 0; c < NS; ++c) S[c] += dpp_shl0<4>(S[c]);
#pragma unroll
    for (int c = 0; c < NS; ++c) S[c] += dpp_shl0<8>(S[c]);
    constexpr int NROWS = (NP + 15) / 16;
#pragma unroll
    for (int r = NROWS - 2; r >= 0; --r) {
        const bool in = (lane >> 4) == r;
#pragma unroll
        for (int c = 0; c < NS; ++c) {
            const double t = readlane_d(S[c], 16 * (r + 1));
            S[c] += in ? t : 0.0;
        }
    }
}

// ----------------------------------------------------------------------------- ground contact
//
// ForceGroundCuboid.computeValues_ / computeEnergy_ (matlab-diff/+redmax/ForceGroundCuboid.m:54-183): penalty contact of the
// 8 corners of a cuboid with the plane (xg, n): normal spring-damper, static / dynamic friction branch per corner.  The
// reference builds body-frame blocks fm, Km, Dm and pulls them through J; conjugating its formulas by R / Ad (DESIGN.md,
// tests/proto_worldframe.py contact_world) gives, per penetrating corner x = R xl + p with velocity vw = v_O + w x x,
// d = n.(x - xg) <= 0 and Gw = [-[x], I]:
//     F  += Gw' f       f = -kn d n - kd N vw   [ - kt T vw  |  - mu kn d t ]
//     Kw += Gw' [XL XR] XL = -kn (d[n] - N[x]) - kd ([N vw] - N[vw])  [ - kt ([a] - T[vw]) | - mu kn (d[t] - d AT [vw] - t (n x x)') ]
//                       XR = -kn N                                    [                    | - mu kn t n' ]
//     Dw += Gw' Y Gw    Y  = -kd N   [ - kt T  |  - mu kn d AT ]      a = T vw, t = a/|a|, AT = (|a|^2 I - a a')/|a|^3 - N/|a|
// as world-frame tensors, so nothing is rotated back and forth.  DERIV=false: wrench and energy only (line-search points).
__device__ __forceinline__ void skew3(const double a[3], double S[9]) {
    S[0] = 0.0;   S[1] = -a[2]; S[2] = a[1];
    S[3] = a[2];  S[4] = 0.0;   S[5] = -a[0];
    S[6] = -a[1]; S[7] = a[0];  S[8] = 0.0;
}
// out(6x6 row-major) += Gw' [CL CR] = [[x] C; C] for the 3x6 block C = [CL CR]
__device__ __forceinline__ void acc_GwT(const double x[3], const double CL[9], const double CR[9], double (&out)[36]) {
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        const double* Cm = c < 3 ? CL : CR;
        const int cc = c < 3 ? c : c - 3;
        const double col[3] = {Cm[cc], Cm[3 + cc], Cm[6 + cc]};
        double xc[3];
        cross3(x, col, xc);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            out[6 * r + c] += xc[r];
            out[6 * (3 + r) + c] += col[r];
        }
    }
}

// index of entry (r, c), r <= c, of a symmetric 6x6 matrix stored as its 21 upper-triangle entries, rows first
__host__ __device__ constexpr int sym21(const int r, const int c) { return r * 6 - r * (r - 1) / 2 + (c - r); }
// MODE 0: wrench and energy only (F, V: the front, every evaluation).  MODE 1: the stiffness block Kw (36, row-major).
// MODE 2: the damping block Dw = sum Gw' Y Gw, symmetric because N, T and AT are: its 21 upper-triangle entries, rows first
// (00..05, 11..15, 22..25, 33..35, 44, 45, 55).  The two derivative passes each repeat the corner geometry; keeping both blocks
// (72 accumulators) live at once is what pushed the Hessian stage of the contact kernels into scratch.
// The ground frame and constants of this lane's ForceGroundCuboid object (every object holds its own E, kn, kt, mu, kd:
// ForceGroundCuboid.m:6-13), from the rows con_setup stages behind the per-node constants.
struct GroundC {
    double n[3], gx[3], kn, kt, mu, kdc;
};
template <int NP>
__device__ __forceinline__ GroundC ground_of(const DevModel& M, const double* __restrict__ sAcc, const int jc) {
    constexpr int CS = cstride(NP);
    const double* g = RMX_CONSTS(sAcc, M.n, NP) + NCONST * CS + jc;
    GroundC G;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        G.n[c] = g[c * CS];
        G.gx[c] = g[(3 + c) * CS];
    }
    G.kn = g[6 * CS];
    G.kt = g[7 * CS];
    G.mu = g[8 * CS];
    G.kdc = g[9 * CS];
    return G;
}
template <int MODE>
__device__ __forceinline__ bool contact_body(const GroundC& G, const bool con, const double sd[3], const double R[9],
                                             const double p[3], const double phw[3], const double phv[3], double (&F)[6],
                                             double (&KD)[36], double& V, bool* lane_pen = nullptr) {
    const double n[3] = {G.n[0], G.n[1], G.n[2]};
    const double kn = G.kn, kt = G.kt, mu = G.mu, kdc = G.kdc;
    if (MODE == 0) {
#pragma unroll
        for (int c = 0; c < 6; ++c) F[c] = 0.0;
        V = 0.0;
    } else {
#pragma unroll
        for (int c = 0; c < (MODE == 1 ? 36 : 21); ++c) KD[c] = 0.0;
    }
    // Which of the 8 corners (+-sides/2, ForceGroundCuboid.m:71-83, the reference's order: ic = 4 [x > 0] + 2 [y > 0] + [z > 0]) of THIS
    // lane's body penetrate (:84-88).  The loop below then visits, per lane, only its own penetrating corners, lowest ic first: the
    // trip count is the largest number of penetrating corners of any body of the tree (4 for a cuboid lying on a face, 2 on an edge)
    // instead of the 8 that the union over 32 bodies in different orientations always came to - and every body still adds its
    // corners in the reference's order, so the sums are the same bit for bit.
    unsigned rem = 0u;
#pragma unroll
    for (int ic = 0; ic < 8; ++ic) {
        const double xl[3] = {(ic & 4 ? 0.5 : -0.5) * sd[0], (ic & 2 ? 0.5 : -0.5) * sd[1], (ic & 1 ? 0.5 : -0.5) * sd[2]};
        double x[3];
        mat3v(R, xl, x);
#pragma unroll
        for (int c = 0; c < 3; ++c) x[c] += p[c];
        const double d = n[0] * (x[0] - G.gx[0]) + n[1] * (x[1] - G.gx[1]) + n[2] * (x[2] - G.gx[2]);
        if (con && !(d > 0.0)) rem |= 1u << ic;
    }
    if (lane_pen) *lane_pen = rem != 0u;      // this lane's body has a corner in the ground (eval_front_pair: which of two iterates)
    bool touched = false;
#pragma unroll 1     // unrolled, the scheduler interleaves the corners and their temporaries spill
    while (__any(rem != 0u)) {
        const bool pen = rem != 0u;
        const int ic = pen ? __builtin_ctz(rem) : 0;       // this lane's next penetrating corner
        rem &= rem - 1u;
        const double xl[3] = {(ic & 4 ? 0.5 : -0.5) * sd[0], (ic & 2 ? 0.5 : -0.5) * sd[1], (ic & 1 ? 0.5 : -0.5) * sd[2]};
        double x[3];
        mat3v(R, xl, x);
#pragma unroll
        for (int c = 0; c < 3; ++c) x[c] += p[c];
        const double d = n[0] * (x[0] - G.gx[0]) + n[1] * (x[1] - G.gx[1]) + n[2] * (x[2] - G.gx[2]);
        touched = true;
        if (pen) {
            if (MODE == 0) V += 0.5 * kn * d * d;    // (:176)
            double vw[3], t3[3];
            cross3(phw, x, t3);
#pragma unroll
            for (int c = 0; c < 3; ++c) vw[c] = phv[c] + t3[c];
            const double nv = dot3(n, vw);
            double a[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) a[c] = vw[c] - n[c] * nv;            // T vw
            double an, ian;
            sqrt_rcp(dot3(a, a), an, ian);
            const bool fric = mu != 0.0;
            const bool stat = fric && (mu * fabs(kn * d) > kt * an);   // (:112)
            const double mukn = mu * kn;
            double tt[3] = {0.0, 0.0, 0.0};
            if (fric && !stat) {
                const double ia = ian;
#pragma unroll
                for (int c = 0; c < 3; ++c) tt[c] = a[c] * ia;
            }
            if (MODE == 0) {
                double f[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    f[c] = -kn * d * n[c] - kdc * nv * n[c];
                    if (fric) f[c] -= stat ? kt * a[c] : mukn * d * tt[c];
                }
                cross3(x, f, t3);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    F[c] += t3[c];
                    F[3 + c] += f[c];
                }
            }
            if (MODE == 1) {
                double nx[3], nxv[3];
                cross3(n, x, nx);
                cross3(n, vw, nxv);
                double XL[9], XR[9], Sk[9];
                // normal spring + damper
                const double Nv[3] = {n[0] * nv, n[1] * nv, n[2] * nv};
                double Sn[9], SNv[9];
                skew3(n, Sn);
                skew3(Nv, SNv);
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        XL[3 * i + k] = -kn * (d * Sn[3 * i + k] - n[i] * nx[k]) - kdc * (SNv[3 * i + k] - n[i] * nxv[k]);
                        XR[3 * i + k] = -kn * (n[i] * n[k]);
                    }
                if (fric) {
                    double Sv[9];
                    skew3(vw, Sv);
                    if (stat) {
                        skew3(a, Sk);
#pragma unroll
                        for (int i = 0; i < 3; ++i)
#pragma unroll
                            for (int k = 0; k < 3; ++k)
                                XL[3 * i + k] -= kt * (Sk[3 * i + k] - (Sv[3 * i + k] - n[i] * nxv[k]));
                    } else {
                        const double ia = ian, ia3 = ia * ia * ia, a2 = an * an;
                        double AT[9], ATS[9];
#pragma unroll
                        for (int i = 0; i < 3; ++i)
#pragma unroll
                            for (int k = 0; k < 3; ++k)
                                AT[3 * i + k] = ((i == k ? a2 : 0.0) - a[i] * a[k]) * ia3 - n[i] * n[k] * ia;
#pragma unroll
                        for (int i = 0; i < 3; ++i)
#pragma unroll
                            for (int k = 0; k < 3; ++k)
                                ATS[3 * i + k] = AT[3 * i] * Sv[k] + AT[3 * i + 1] * Sv[3 + k] + AT[3 * i + 2] * Sv[6 + k];
                        skew3(tt, Sk);
#pragma unroll
                        for (int i = 0; i < 3; ++i)
#pragma unroll
                            for (int k = 0; k < 3; ++k) {
                                XL[3 * i + k] -= mukn * (d * Sk[3 * i + k] - d * ATS[3 * i + k] - tt[i] * nx[k]);
                                XR[3 * i + k] -= mukn * tt[i] * n[k];
                            }
                    }
                }
                acc_GwT(x, XL, XR, KD);
            }
            if (MODE == 2) {
                double Y[9];
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int k = 0; k < 3; ++k) Y[3 * i + k] = -kdc * (n[i] * n[k]);
                if (fric) {
                    if (stat) {
#pragma unroll
                        for (int i = 0; i < 3; ++i)
#pragma unroll
                            for (int k = 0; k < 3; ++k) Y[3 * i + k] -= kt * ((i == k ? 1.0 : 0.0) - n[i] * n[k]);
                    } else {
                        const double ia = ian, ia3 = ia * ia * ia, a2 = an * an;
#pragma unroll
                        for (int i = 0; i < 3; ++i)
#pragma unroll
                            for (int k = 0; k < 3; ++k)
                                Y[3 * i + k] -= mukn * d * (((i == k ? a2 : 0.0) - a[i] * a[k]) * ia3 - n[i] * n[k] * ia);
                    }
                }
                // Gw' Y Gw = [[x] Y [x]' , [x] Y ; Y [x]' , Y]  ([x]' = -[x]); YL = -Y [x]
                double Sx[9], YL[9], TT[9];
                skew3(x, Sx);
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int k = 0; k < 3; ++k)
                        YL[3 * i + k] = -(Y[3 * i] * Sx[k] + Y[3 * i + 1] * Sx[3 + k] + Y[3 * i + 2] * Sx[6 + k]);
                // top rows: [x] [YL Y]
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const double cl[3] = {YL[k], YL[3 + k], YL[6 + k]}, cr[3] = {Y[k], Y[3 + k], Y[6 + k]};
                    double xl3[3], xr3[3];
                    cross3(x, cl, xl3);
                    cross3(x, cr, xr3);
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        TT[3 * i + k] = xl3[i];          // ([x] YL)(i,k)
                        if (true) {
                            // ([x] Y)(i,k): rows 0..2, columns 3..5
                            const int r = i, c = 3 + k;
                            KD[sym21(r, c)] += xr3[i];
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int k = i; k < 3; ++k) {
                        KD[sym21(i, k)] += TT[3 * i + k];
                        KD[sym21(3 + i, 3 + k)] += Y[3 * i + k];
                    }
            }
        }
    }
    return touched;
}

// Subtree sums of NC <= 28 per-node numbers through the LDS transpose (the scan of eval_front_e2 as a function: contact's 72
// extra accumulations go through it in chunks).  Row n of sAcc is zero; ends with a barrier so sAcc can be rewritten.
template <int NP, int NC>
__device__ __forceinline__ void lds_subtree_sum(const DevModel& M, double* __restrict__ sAcc, const double* cEnd, const int lane,
                                                const bool act, const int jj, double* v) {
    static_assert(NC <= NACC, "chunk wider than the accumulation row");
    const int n = M.n;
    if (act) {
        double* A = sAcc + lane * ACC_STRIDE;
#pragma unroll
        for (int c = 0; c < NC; ++c) A[c] = v[c];
    }
    RMX_SYNC();
    if constexpr (NP >= 32) {
        // two lanes per component (see eval_front_e2): lane c scans the upper half of the nodes, lane 32+c the lower half
        constexpr int HALF = NP / 2;
        const int comp = lane & 31;
        const int base = lane >= 32 ? 0 : HALF;
        const bool on = comp < NC;
        double a[HALF];
        const bool full = n == NP;             // wave-uniform: no per-row bounds needed
        if (full) {
#pragma unroll
            for (int t = 0; t < HALF; ++t) a[t] = sAcc[(base + t) * ACC_STRIDE + comp];
        } else {
#pragma unroll
            for (int t = 0; t < HALF; ++t) a[t] = (on && base + t < n) ? sAcc[(base + t) * ACC_STRIDE + comp] : 0.0;
        }
        double acc = 0.0;
#pragma unroll
        for (int t = HALF - 1; t >= 0; --t) {
            acc += a[t];
            a[t] = acc;
        }
        const double tail = dup_lo(acc);
        if (lane >= 32) {
#pragma unroll
            for (int t = 0; t < HALF; ++t) a[t] += tail;
        }
        if (on) {
            if (full) {
#pragma unroll
                for (int t = 0; t < HALF; ++t) sAcc[(base + t) * ACC_STRIDE + comp] = a[t];
            } else {
#pragma unroll
                for (int t = 0; t < HALF; ++t)
                    if (base + t < n) sAcc[(base + t) * ACC_STRIDE + comp] = a[t];
            }
        }
    } else if (lane < NC) {
        double a[NP];
#pragma unroll
        for (int jn = 0; jn < NP; ++jn) a[jn] = (jn < n) ? sAcc[jn * ACC_STRIDE + lane] : 0.0;
        double acc = 0.0;
#pragma unroll
        for (int jn = NP - 1; jn >= 0; --jn) {
            acc += a[jn];
            a[jn] = acc;
        }
#pragma unroll
        for (int jn = 0; jn < NP; ++jn)
            if (jn < n) sAcc[jn * ACC_STRIDE + lane] = a[jn];
    }
    RMX_SYNC();
    {
        const double* A = sAcc + jj * ACC_STRIDE;
        const int en = (act && !M.is_chain) ? (int)cEnd[jj] : n;
        const double* E = sAcc + en * ACC_STRIDE;
#pragma unroll
        for (int c = 0; c < NC; ++c) v[c] = A[c] - E[c];
    }
    RMX_SYNC();
}

// FULL: accumulate the Hessian's subtree sums as well (28 numbers per body instead of 6).
// e2 is the coefficient of f in g = M v - e2 f: eta^2 for the implicit integrators; the linearly-implicit Euler step of
// matlab-simple uses e2 = -h with v = qdot0 so that g = M qdot0 + h f is its right-hand side.
// AS: the row stride of the accumulation scratch (ACC_STRIDE; a residual-only evaluation on a scratch of its own may take a smaller odd one)
template <int NP, bool FULL, bool TIMED = false, bool CT = false, bool NEARCHK = false, int AS = ACC_STRIDE>
__device__ __forceinline__ void eval_front_e2(const DevModel& M, double* __restrict__ sAcc, const int lane, const double xq,
                                              const double xqd, const double xv, const double eta, const double e2, NodeOut& out,
                                              FrontState& fs, unsigned long long* stamps = nullptr) {
    unsigned long long last_ = TIMED ? __builtin_amdgcn_s_memtime() : 0ull;
    const int n = M.n;
    const bool act = lane < n;
    const int jj = act ? lane : 0;
    // per-node constants staged in LDS by smem_setup (stride NP): reading them from L2 at every evaluation left the single
    // resident wave parked at s_waitcnt (SQ_WAIT_ANY 31 % of wave cycles)
    constexpr int CS = cstride(NP);
    const int jc = (CS > NP && lane >= NP) ? NP : lane;      // this lane's column of constants (idle column beyond NP)
    const double* cK = RMX_CONSTS(sAcc, n, NP);
    const double* cSb = cK + 36 * CS;
    const double* cI4 = cSb + 6 * CS;
    const double* cPrm = cI4 + 4 * CS;
    const double* cTyp = cPrm + 8 * CS;     // joint type, stored as a double
    const double* cRel = cTyp + CS;         // 2 rows: ancestor / descendant bit masks (bit patterns)
    const double* cAnc = cRel + 2 * CS;     // MAXROUNDS rows: ancestor 2^r levels up (as doubles), trees only
    const double* cEnd = cAnc + MAXROUNDS * CS;
    const double* cCon = cEnd + CS;         // 4 rows: contact flag, cuboid sides
    const int type = (int)cTyp[jc];
    const bool dof = type != 0;
    fs.anc_m = (unsigned long long)__double_as_longlong(cRel[jc]);
    fs.desc_m = (unsigned long long)__double_as_longlong(cRel[CS + jc]);

    // callers pass zeros on lanes without a DOF (fixed joints, idle lanes): no selects needed here
    const double q = xq, qd = xqd, v = xv;

    // ---- joint transform T_j(q) = K0 + u K1 + w K2  (JointRevolute/Prismatic.update_, Joint.update :401-408,
    //      Body.update :70-72, folded with the constant offsets E0_ij(parent) E0_pj and E0_ji)
    double u = 0.0, w = 0.0;
    if (type == 1) {
        sincos(q, &u, &w);
    } else if (type == 2) {
        u = q;
    }
    double R[9], p[3];
#pragma unroll
    for (int c = 0; c < 9; ++c) R[c] = cK[c * CS + jc] + u * cK[(12 + c) * CS + jc] + w * cK[(24 + c) * CS + jc];
#pragma unroll
    for (int c = 0; c < 3; ++c) p[c] = cK[(9 + c) * CS + jc] + u * cK[(21 + c) * CS + jc] + w * cK[(33 + c) * CS + jc];
    // idle lanes read the identity transform from their constants, so they are neutral in the chain scans

    RMX_STAMP(0)
    // ---- world transforms E_w,j = E_w,parent T_j
    if (M.is_chain) {
        chain_scan_transform<NP>(lane, R, p);
    } else {   // general tree: pointer jumping over ancestors (log2(depth) rounds of cross-lane permutes)
        for (int r = 0; r < M.rounds; ++r) {
            const int a = (int)cAnc[r * CS + jc];
            const int src = a >= 0 ? a : lane;
            double Ra[9], pa[3];
#pragma unroll
            for (int c = 0; c < 9; ++c) Ra[c] = shfl_d(R[c], src);
#pragma unroll
            for (int c = 0; c < 3; ++c) pa[c] = shfl_d(p[c], src);
            if (a >= 0) {
                double Rn[9], pn[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) Rn[3 * i + k] = Ra[3 * i] * R[k] + Ra[3 * i + 1] * R[3 + k] + Ra[3 * i + 2] * R[6 + k];
                    pn[i] = Ra[3 * i] * p[0] + Ra[3 * i + 1] * p[1] + Ra[3 * i + 2] * p[2] + pa[i];
                }
#pragma unroll
                for (int c = 0; c < 9; ++c) R[c] = Rn[c];
#pragma unroll
                for (int c = 0; c < 3; ++c) p[c] = pn[c];
            }
        }
    }

    RMX_STAMP(1)
    // ---- world-frame joint screw s_j = Ad(E_w,j) (A0_ij S)   (the column of J, Joint.m:508-522)
    double sbw[3], sbv[3], t3[3];
    double (&sw)[3] = fs.sw;
    double (&sv)[3] = fs.sv;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        sbw[c] = cSb[c * CS + jc];
        sbv[c] = cSb[(3 + c) * CS + jc];
    }
    mat3v(R, sbw, sw);
    mat3v(R, sbv, sv);
    cross3(p, sw, t3);
#pragma unroll
    for (int c = 0; c < 3; ++c) sv[c] += t3[c];

    // ---- phi_j = sum_{a in anc*(j)} s_a qdot_a   ( = (J qdot)_j, Joint.update :411-419 )
    double (&phw)[3] = fs.phw;
    double (&phv)[3] = fs.phv;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        phw[c] = sw[c] * qd;
        phv[c] = sv[c] * qd;
    }
    if (M.is_chain) {
        chain_scan_sum6<NP>(lane, phw, phv);
    } else {
        for (int r = 0; r < M.rounds; ++r) {
            const int a = (int)cAnc[r * CS + jc];
            const int src = a >= 0 ? a : lane;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const double tw = shfl_d(phw[c], src), tv = shfl_d(phv[c], src);
                if (a >= 0) {
                    phw[c] += tw;
                    phv[c] += tv;
                }
            }
        }
    }
    RMX_STAMP(2)
    // ---- xi_j = ad(phi_j) s_j ; beta_j = sum_{a in anc*(j)} (s_a v_a + eta^2 xi_a qdot_a)  ( = (J v + eta^2 Jdot qdot)_j )
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
    if (M.is_chain) {
        chain_scan_sum6<NP>(lane, bw, bv);
    } else {
        for (int r = 0; r < M.rounds; ++r) {
            const int a = (int)cAnc[r * CS + jc];
            const int src = a >= 0 ? a : lane;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const double tw = shfl_d(bw[c], src), tv = shfl_d(bv[c], src);
                if (a >= 0) {
                    bw[c] += tw;
                    bv[c] += tv;
                }
            }
        }
    }

    RMX_STAMP(3)
    // ---- world-frame spatial inertia of body j (Body.computeMassGrav :99-101): m, mc, Ibar = R diag(I) R' + m [c][c]'
    const double I1 = cI4[0 * CS + jc], I2 = cI4[1 * CS + jc];
    const double I3 = cI4[2 * CS + jc], ms = cI4[3 * CS + jc];
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
    // momentum h = I phi, I beta
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
    // Coriolis wrench ad(phi)' h (Body.m:102-103) and gravity wrench (Body.m:104-109), world frame
    double fct[3], fcf[3], a3[3], b3[3];
    cross3(phw, ht, a3);
    cross3(phv, hf, b3);
    cross3(phw, hf, fcf);
    const double gv[3] = {M.grav[0], M.grav[1], M.grav[2]};
    double fgt[3];
    cross3(mc, gv, fgt);
    double wt[3], wf[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        fct[c] = -a3[c] - b3[c];
        wt[c] = bt[c] - e2 * (fct[c] + fgt[c]);
        wf[c] = bf[c] - e2 * (-fcf[c] + ms * gv[c]);
    }
    // ground contact wrench of this body, world frame (its K/D blocks are formed by eval_hess, once per Newton iteration)
    double eVc = 0.0;
    fs.touched = false;
    if constexpr (CT || NEARCHK) {
        // The lowest corner of the cuboid sits  n.(p - xg) - 1/2 sum_i |n.R_i| sides_i  above the plane (R_i: column i of the
        // body's rotation).  While that is positive for every body of the tree (wave-uniform, with a margin far above the
        // rounding of either form) no corner penetrates: the corner loop would find nothing and leave Fc = 0, eVc = 0,
        // touched = false, and it is skipped.  NEARCHK (the lean Newton of the contact-capable step kernels, newton_node):
        // only this test runs and its outcome is reported in fs.touched; the caller leaves the lean path when it is set.
        const bool con = cCon[jc] != 0.0;
        const double sd[3] = {cCon[CS + jc], cCon[2 * CS + jc], cCon[3 * CS + jc]};
        const GroundC G = ground_of<NP>(M, sAcc, jc);
        const double dc = G.n[0] * (p[0] - G.gx[0]) + G.n[1] * (p[1] - G.gx[1]) + G.n[2] * (p[2] - G.gx[2]);
        const double reach = 0.5 * (fabs(G.n[0] * R[0] + G.n[1] * R[3] + G.n[2] * R[6]) * sd[0] +
                                    fabs(G.n[0] * R[1] + G.n[1] * R[4] + G.n[2] * R[7]) * sd[1] +
                                    fabs(G.n[0] * R[2] + G.n[1] * R[5] + G.n[2] * R[8]) * sd[2]);
        const bool near = __any(con && !(dc - reach > 1e-9 * (fabs(dc) + reach)));
        if constexpr (!CT) {
            fs.touched = near;
        } else if (near) {
            double Fc[6], k1[36];
            fs.touched = contact_body<0>(G, con, sd, R, p, phw, phv, Fc, k1, eVc);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                wt[c] -= e2 * Fc[c];
                wf[c] -= e2 * Fc[3 + c];
            }
        }
    }

    // energies (Body.computeEnergies Body.m:167-173, Joint.computeEnergies Joint.m:616-637)
    const double stiff = cPrm[1 * CS + jc], damp = cPrm[2 * CS + jc];
    const double tau = cPrm[0 * CS + jc], qRest = cPrm[3 * CS + jc];
    const double qLimL = cPrm[4 * CS + jc], qLimU = cPrm[5 * CS + jc];
    const double qLimK = cPrm[6 * CS + jc], qLimD = cPrm[7 * CS + jc];
    const double hitL = (dof && q < qLimL) ? 1.0 : 0.0, hitU = (dof && q > qLimU) ? 1.0 : 0.0;
    {
        double eT = 0.5 * (dot3(phw, ht) + dot3(phv, hf));
        double eV = -dot3(gv, mc);
        if (dof) {
            const double dq = q - qRest;
            const double dqL = hitL * (qLimL - q), dqU = hitU * (qLimU - q);
            eV += 0.5 * stiff * (dq * dq) + 0.5 * qLimK * (dqL * dqL + dqU * dqU);
        }
        out.eT = eT;              // idle lanes: zero mass, zero twist, no joint -> exact zeros
        out.eV = eV + eVc;
    }

    RMX_STAMP(4)
    // ---- subtree sums: W (6) [+ m, mc, Ibar, TL, hf for the Hessian].  Idle lanes hold zeros (zero mass).
    constexpr int NS = FULL ? NACC : 6;
    double (&S)[NACC] = fs.S;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        S[c] = wt[c];
        S[3 + c] = wf[c];
    }
    if (FULL) {
        S[6] = ms;
#pragma unroll
        for (int c = 0; c < 3; ++c) S[7 + c] = mc[c];
#pragma unroll
        for (int c = 0; c < 6; ++c) S[10 + c] = Ib[c];
        // TL = X + X' + [h_tau],  X = Ibar [phi_w] + [mc][phi_v]   (B = I ad(phi) + ad(phi)' I + N(h) = [[TL,0],[2[hf],0]])
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
#pragma unroll
        for (int c = 0; c < 3; ++c) S[25 + c] = hf[c];
    }
    RMX_STAMP(5)
    if (M.is_chain && !FULL) {
        // measured: the register (DPP) scan wins for the 6-number residual-only accumulation (10.5k vs 12.3k cycles per
        // evaluation) but loses for the full 28-number one (24.7k vs 23.9k), which stays on the LDS transpose below
        chain_suffix_sum<NP, NS>(lane, S);
        RMX_STAMP(6)
    } else {
        // general tree: transpose through LDS (stride 29: conflict-free), one lane per component scans the nodes in
        // depth-first order in registers; subtree(j) = suffix(j) - suffix(end_j), row n of sAcc is kept zero
        if (act) {
            double* A = sAcc + lane * AS;
#pragma unroll
            for (int c = 0; c < NS; ++c) A[c] = S[c];
        }
        RMX_SYNC();
        RMX_STAMP(5)
        if constexpr (NP == 32 || NP == 64) {
            // the scan needs NS <= 28 lanes, one per component: every component gets TWO lanes instead, lane c over the upper
            // half of the nodes and lane 32+c over the lower half, which then adds the upper half's total
            // (v_permlane32_swap): half the serial chain, half the LDS operations per lane
            constexpr int HALF = NP / 2;
            const int comp = lane & 31;
            const int base = lane >= 32 ? 0 : HALF;
            const bool on = comp < NS;
            double a[HALF];
            const bool full = n == NP;             // wave-uniform: every node slot in use, no per-row bounds needed
            if (full) {
#pragma unroll
                for (int t = 0; t < HALF; ++t) a[t] = sAcc[(base + t) * AS + comp];   // lanes >= NS read finite junk, never stored
            } else {
#pragma unroll
                for (int t = 0; t < HALF; ++t) a[t] = (on && base + t < n) ? sAcc[(base + t) * AS + comp] : 0.0;
            }
            double acc = 0.0;
#pragma unroll
            for (int t = HALF - 1; t >= 0; --t) {
                acc += a[t];
                a[t] = acc;
            }
            const double tail = dup_lo(acc);           // lanes >= 32: the sum over the upper half of the nodes
            if (lane >= 32) {
#pragma unroll
                for (int t = 0; t < HALF; ++t) a[t] += tail;
            }
            if (on) {
                if (full) {
#pragma unroll
                    for (int t = 0; t < HALF; ++t) sAcc[(base + t) * AS + comp] = a[t];
                } else {
#pragma unroll
                    for (int t = 0; t < HALF; ++t)
                        if (base + t < n) sAcc[(base + t) * AS + comp] = a[t];
                }
            }
        } else if (lane < NS) {
            double a[NP];
#pragma unroll
            for (int jn = 0; jn < NP; ++jn) a[jn] = (jn < n) ? sAcc[jn * AS + lane] : 0.0;
            double acc = 0.0;
#pragma unroll
            for (int jn = NP - 1; jn >= 0; --jn) {
                acc += a[jn];
                a[jn] = acc;
            }
#pragma unroll
            for (int jn = 0; jn < NP; ++jn)
                if (jn < n) sAcc[jn * AS + lane] = a[jn];
        }
        RMX_STAMP(6)
        RMX_SYNC();
        {
            const double* A = sAcc + jj * AS;
#pragma unroll
            for (int c = 0; c < NS; ++c) S[c] = A[c];
            if (!M.is_chain) {
                const int en = (int)cEnd[jc];
                const double* E = sAcc + en * AS;
                // (every load in flight before the first subtraction: written as S[c] -= E[c] the compiler gives the loads ONE
                // destination register and a full wait each - 28 LDS round trips, 1.5 k cycles per evaluation of a branching tree)
                double ev[NS];
#pragma unroll
                for (int c = 0; c < NS; ++c) ev[c] = E[c];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < NS; ++c) S[c] -= ev[c];
            }
        }
        RMX_SYNC();   // sAcc is rewritten by the next evaluation
    }
    RMX_STAMP(7)
    // ---- residual  g_j = s_j . W_j - eta^2 fr_j   (Joint.computeForce Joint.m:437-456, evalBDF1 :180)
    const double fr = (tau + fs.tau_add) + stiff * (qRest - q) - damp * qd + hitL * (qLimK * (qLimL - q) - qLimD * qd) +
                      hitU * (qLimK * (qLimU - q) - qLimD * qd);
    out.g = dof ? (dot3(sw, &S[0]) + dot3(sv, &S[3]) - e2 * fr) : 0.0;
#pragma unroll
    for (int c = 0; c < 9; ++c) fs.Rw[c] = R[c];
#pragma unroll
    for (int c = 0; c < 3; ++c) fs.pw[c] = p[c];
    fs.eta = eta;
    fs.kd = stiff + (hitL + hitU) * qLimK;    // -Kr  (Joint.computeForce Joint.m:470-482)
    fs.dd = damp + (hitL + hitU) * qLimD;     // -Dr
    fs.act = act;
    fs.dof = dof;
    RMX_STAMP(8)
}

template <int NP, bool FULL, bool TIMED = false, bool CT = false, bool NEARCHK = false>
__device__ __forceinline__ void eval_front(const DevModel& M, double* __restrict__ sAcc, const int lane, const double xq,
                                           const double xqd, const double xv, const double eta, NodeOut& out, FrontState& fs,
                                           unsigned long long* stamps = nullptr) {
    eval_front_e2<NP, FULL, TIMED, CT, NEARCHK>(M, sAcc, lane, xq, xqd, xv, eta, eta * eta, out, fs, stamps);
}

// ----------------------------------------------------------------------------- two line-search points per evaluation (n <= 32)
//
// A tree of at most 32 nodes leaves lanes 32..63 idle in every lane = node stage of the front.  The trial points of newton()'s
// backtracking line search (driverRedMaxBDF1.m:124-138: x0 + alpha dx for alpha = 1, 1/2, 1/4 ... until 0.5 |g|^2 decreases, at most
// iterLsMax of them) do not depend on each other, only the DECISION which one is taken does.  eval_front_dual evaluates the residual
// at TWO iterates in one pass: lanes 0..31 carry trial point a, lanes 32..63 trial point b = the next halving, node = lane & 31 in
// both halves, for the instruction count of one evaluation (every stage is lane-local or stays inside 16-lane DPP rows; the
// hand-over from the first to the second row of a half is one row_bcast:15 for both halves at once).  The caller takes a if it is
// accepted, else b, else goes on with the next pair: the reference's decisions in the reference's order, half the evaluations.
// Residual only (6 subtree sums, by the register scan): the accepted point is evaluated once more by the full front, whose state
// the Hessian stage needs.  Serial chains only (the callers fall back to one point per evaluation for branching trees).
// What it is for: BASELINE.json configs[4] - at a stick / slip kink of the ground contact the reference's line search runs out its
// 20 trials (or accepts 2^-13 of the step) on every one of the 320 iterations of a step, ~4300 trial evaluations for one step of one
// rollout, and the launch of the whole batch waits for it.
//
// lane 15 of every 16-lane row -> all lanes of the NEXT row; row 0 receives 0 (bound_ctrl).  No row mask: the destination needs no
// pre-loaded fill value (two v_mov per double in the masked form); rows 0 and 2 receive what their callers ignore - a 0 / the other
// chain's finite total under a 0 weight, or operands of a composition only rows 1 and 3 perform
__device__ __forceinline__ double dpp_bcast15_rows(const double v) {
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x142, 0xF, 0xF, true);
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x142, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
// inclusive prefix sums along the chain, two chains of 32 nodes side by side (lanes 0..31 and 32..63)
__device__ __forceinline__ void chain_scan_sum6_dual(const int lane, double (&a)[3], double (&b)[3]) {
#pragma unroll
    for (int c = 0; c < 3; ++c) { a[c] += dpp_shr0<1>(a[c]); b[c] += dpp_shr0<1>(b[c]); }
#pragma unroll
    for (int c = 0; c < 3; ++c) { a[c] += dpp_shr0<2>(a[c]); b[c] += dpp_shr0<2>(b[c]); }
#pragma unroll
    for (int c = 0; c < 3; ++c) { a[c] += dpp_shr0<4>(a[c]); b[c] += dpp_shr0<4>(b[c]); }
#pragma unroll
    for (int c = 0; c < 3; ++c) { a[c] += dpp_shr0<8>(a[c]); b[c] += dpp_shr0<8>(b[c]); }
    // rows 1 and 3 add the complete total of lane 15 / 47: one FMA with a 0/1 lane weight (exact: 1 t + a, 0 t + a), as
    // chain_scan_sum6 hands its row totals over
    const double w = (lane & 16) ? 1.0 : 0.0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        a[c] = fma(w, dpp_bcast15_rows(a[c]), a[c]);
        b[c] = fma(w, dpp_bcast15_rows(b[c]), b[c]);
    }
}
__device__ __forceinline__ void chain_scan_transform_dual(const int lane, double (&R)[9], double (&p)[3]) {
    chain_compose_step<1>(lane, R, p);
    chain_compose_step<2>(lane, R, p);
    chain_compose_step<4>(lane, R, p);
    chain_compose_step<8>(lane, R, p);
    double Ra[9], pa[3];
#pragma unroll
    for (int c = 0; c < 9; ++c) Ra[c] = dpp_bcast15_rows(R[c]);
#pragma unroll
    for (int c = 0; c < 3; ++c) pa[c] = dpp_bcast15_rows(p[c]);
    if (lane & 16) {
        double Rn[9], pn[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int k = 0; k < 3; ++k) Rn[3 * i + k] = Ra[3 * i] * R[k] + Ra[3 * i + 1] * R[3 + k] + Ra[3 * i + 2] * R[6 + k];
            pn[i] = Ra[3 * i] * p[0] + Ra[3 * i + 1] * p[1] + Ra[3 * i + 2] * p[2] + pa[i];
        }
#pragma unroll
        for (int c = 0; c < 9; ++c) R[c] = Rn[c];
#pragma unroll
        for (int c = 0; c < 3; ++c) p[c] = pn[c];
    }
}
// suffix sums of 6 numbers per node along the two chains: row_shl scans, then rows 0 and 2 add the total of lane 16 / 48
__device__ __forceinline__ void chain_suffix_sum6_dual(const int lane, double (&S)[6]) {
#pragma unroll
    for (int c = 0; c < 6; ++c) S[c] += dpp_shl0<1>(S[c]);
#pragma unroll
    for (int c = 0; c < 6; ++c) S[c] += dpp_shl0<2>(S[c]);
#pragma unroll
    for (int c = 0; c < 6; ++c) S[c] += dpp_shl0<4>(S[c]);
#pragma unroll
    for (int c = 0; c < 6; ++c) S[c] += dpp_shl0<8>(S[c]);
    const double w0 = ((lane >> 4) == 0) ? 1.0 : 0.0, w2 = ((lane >> 4) == 2) ? 1.0 : 0.0;
    double t0[6], t2[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        t0[c] = readlane_d(S[c], 16);
        t2[c] = readlane_d(S[c], 48);
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) S[c] = fma(w2, t2[c], fma(w0, t0[c], S[c]));
}
// the two halves' sums over their 32 lanes, each bitwise what wave_sum gives for a wave whose other half holds zeros
__device__ __forceinline__ void wave_sum_dual(double v, double& sa, double& sb) {
    v += dpp_d<DPP_XOR1>(v);
    v += dpp_d<DPP_XOR2>(v);
    v += dpp_d<DPP_HALF_MIRROR>(v);
    v += dpp_d<DPP_MIRROR>(v);
    sa = (readlane_d(v, 0) + readlane_d(v, 16)) + 0.0;
    sb = (readlane_d(v, 32) + readlane_d(v, 48)) + 0.0;
}

// g of two iterates of a serial chain of n <= 32 nodes (see above).  xq, xqd, xv: the coordinates of node lane & 31 at trial point
// lane >> 5, zeros where the node has no DOF.  Same formulas, in the same order, as eval_front_e2<32, false, false, CT>.
#ifndef RMX_DUAL_INLINE
#define RMX_DUAL_INLINE 1          // 0: eval_front_dual as an out-of-line function with scalar arguments only (build variants)
#endif
#if RMX_DUAL_INLINE
#define RMX_DUAL_FN __device__ __forceinline__
#else
#define RMX_DUAL_FN __device__ __attribute__((noinline))
#endif
struct Grav3 { double x, y, z; };
template <bool CT>
RMX_DUAL_FN double eval_front_dual(const double* __restrict__ cK, const Grav3 grav, const int lane, const double xq,
                                   const double xqd, const double xv, const double eta, const double tau_add) {
    constexpr int NP = 32;
    constexpr int CS = cstride(NP);
    const double e2 = eta * eta;
    const int jc = lane & 31;
    const double* cSb = cK + 36 * CS;
    const double* cI4 = cSb + 6 * CS;
    const double* cPrm = cI4 + 4 * CS;
    const double* cTyp = cPrm + 8 * CS;
    const double* cCon = cTyp + CS + 2 * CS + MAXROUNDS * CS + CS;
    const int type = (int)cTyp[jc];
    const bool dof = type != 0;
    const double q = xq, qd = xqd, v = xv;
    double u = 0.0, w = 0.0;
    if (type == 1) {
        sincos(q, &u, &w);
    } else if (type == 2) {
        u = q;
    }
    double R[9], p[3];
#pragma unroll
    for (int c = 0; c < 9; ++c) R[c] = cK[c * CS + jc] + u * cK[(12 + c) * CS + jc] + w * cK[(24 + c) * CS + jc];
#pragma unroll
    for (int c = 0; c < 3; ++c) p[c] = cK[(9 + c) * CS + jc] + u * cK[(21 + c) * CS + jc] + w * cK[(33 + c) * CS + jc];
    chain_scan_transform_dual(lane, R, p);
    double sbw[3], sbv[3], t3[3], sw[3], sv[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        sbw[c] = cSb[c * CS + jc];
        sbv[c] = cSb[(3 + c) * CS + jc];
    }
    mat3v(R, sbw, sw);
    mat3v(R, sbv, sv);
    cross3(p, sw, t3);
#pragma unroll
    for (int c = 0; c < 3; ++c) sv[c] += t3[c];
    double phw[3], phv[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        phw[c] = sw[c] * qd;
        phv[c] = sv[c] * qd;
    }
    chain_scan_sum6_dual(lane, phw, phv);
    double xiw[3], xiv[3], bw[3], bv[3];
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
    const double I1 = cI4[0 * CS + jc], I2 = cI4[1 * CS + jc];
    const double I3 = cI4[2 * CS + jc], ms = cI4[3 * CS + jc];
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
    const double gv[3] = {grav.x, grav.y, grav.z};
    double fgt[3];
    cross3(mc, gv, fgt);
    double S[6];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        fct[c] = -a3[c] - b3[c];
        S[c] = bt[c] - e2 * (fct[c] + fgt[c]);
        S[3 + c] = bf[c] - e2 * (-fcf[c] + ms * gv[c]);
    }
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
            double Fc[6], k1[36], eVc;
            contact_body<0>(G, con, sd, R, p, phw, phv, Fc, k1, eVc);
#pragma unroll
            for (int c = 0; c < 6; ++c) S[c] -= e2 * Fc[c];
        }
    }
    chain_suffix_sum6_dual(lane, S);
    const double stiff = cPrm[1 * CS + jc], damp = cPrm[2 * CS + jc];
    const double tau = cPrm[0 * CS + jc], qRest = cPrm[3 * CS + jc];
    const double qLimL = cPrm[4 * CS + jc], qLimU = cPrm[5 * CS + jc];
    const double qLimK = cPrm[6 * CS + jc], qLimD = cPrm[7 * CS + jc];
    const double hitL = (dof && q < qLimL) ? 1.0 : 0.0, hitU = (dof && q > qLimU) ? 1.0 : 0.0;
    const double fr = (tau + tau_add) + stiff * (qRest - q) - damp * qd + hitL * (qLimK * (qLimL - q) - qLimD * qd) +
                      hitU * (qLimK * (qLimU - q) - qLimD * qd);
    return dof ? (dot3(sw, &S[0]) + dot3(sv, &S[3]) - e2 * fr) : 0.0;
}

// Reduced mass matrix row M(a,:) = (J' Mm J)(a,:) of this node from the subtree inertias (computeValues :212;
// matlab-simple euler :86-87): M(a,i) = (Ic_a s_a).s_i if i is an ancestor-or-self of a, s_a.(Ic_i s_i) if a descendant.
template <int NP>
__device__ __forceinline__ void eval_mass(const DevModel& M, const int lane, const FrontState& fs, double (&Mrow)[NP]) {
    const bool act = fs.act;
    const int jj = act ? lane : 0;
    const double mS = fs.S[6];
    const double* mcS = &fs.S[7];
    const double* IbS = &fs.S[10];
    double r1[6], t3[3];
    sym3v(IbS, fs.sw, r1);
    cross3(mcS, fs.sv, t3);
#pragma unroll
    for (int c = 0; c < 3; ++c) r1[c] += t3[c];
    cross3(mcS, fs.sw, t3);
#pragma unroll
    for (int c = 0; c < 3; ++c) r1[3 + c] = mS * fs.sv[c] - t3[c];
    double cv[12];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        cv[c] = act ? fs.sw[c] : 0.0;
        cv[3 + c] = act ? fs.sv[c] : 0.0;
        cv[6 + c] = act ? r1[c] : 0.0;
        cv[9 + c] = act ? r1[3 + c] : 0.0;
    }
    const unsigned long long anc_m = fs.anc_m, desc_m = fs.desc_m;
    const double mdiag = fs.dof ? (fs.sw[0] * r1[0] + fs.sw[1] * r1[1] + fs.sw[2] * r1[2] + fs.sv[0] * r1[3] + fs.sv[1] * r1[4] + fs.sv[2] * r1[5]) : 1.0;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        double Ci[12];
#pragma unroll
        for (int c = 0; c < 12; ++c) Ci[c] = readlane_d(cv[c], i);
        const double lo = r1[0] * Ci[0] + r1[1] * Ci[1] + r1[2] * Ci[2] + r1[3] * Ci[3] + r1[4] * Ci[4] + r1[5] * Ci[5];
        const double up = fs.sw[0] * Ci[6] + fs.sw[1] * Ci[7] + fs.sw[2] * Ci[8] + fs.sv[0] * Ci[9] + fs.sv[1] * Ci[10] + fs.sv[2] * Ci[11];
        const double mu = (double)(unsigned)((desc_m >> i) & 1ull);
        const double ml = (double)(unsigned)((anc_m >> i) & 1ull);
        Mrow[i] = (i == lane) ? mdiag : (mu * up + ml * lo);
    }
}

// M and D = df/dqdot rows of this node (computeValues :212, :227-237), what the adjoint backward sweep needs per step
// (TaskBDF1.m:58-70):  D(a,i) = s_a.(Bc_i s_i - 2 Ic_i xi_i)  a ancestor-or-self of i ;
//                              = (Bc_a' s_a).s_i - 2 (Ic_a s_a).xi_i  a strict descendant ;  + Dr on the diagonal.
// Idle rows/columns: identity in M, zero in D.
// Column I (and on) of M and D for trees of up to 16 nodes (eval_MD's loop with the column broadcast fused into the FMAs).
template <int NP, int I, bool CD = false>
__device__ __forceinline__ void md_columns_dpp(const double (&cv)[24], const double (&sw)[3], const double (&sv)[3], const double (&r1)[6],
                                               const double (&r2w)[3], const double (&r2v)[3], const unsigned long long desc_m,
                                               const unsigned long long anc_m, const int lane, const double mdiag, const double ddiag,
                                               double (&Mrow)[NP], double (&Drow)[NP]) {
    if constexpr (I < NP) {
        double m_lo = 0.0, m_up = 0.0, d_a = 0.0, d_b = 0.0, d_up = 0.0;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            fmadd_rowbcast<I>(m_lo, cv[c], r1[c]);
            fmadd_rowbcast<I>(d_b, cv[6 + c], r1[c]);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {         // accumulators in turn, see hess_columns_dpp
            fmadd_rowbcast<I>(m_up, cv[12 + c], sw[c]);
            fmadd_rowbcast<I>(d_a, cv[c], r2w[c]);
            fmadd_rowbcast<I>(d_up, cv[18 + c], sw[c]);
            fmadd_rowbcast<I>(m_up, cv[15 + c], sv[c]);
            fmadd_rowbcast<I>(d_up, cv[21 + c], sv[c]);
            if constexpr (CD) fmadd_rowbcast<I>(d_a, cv[3 + c], r2v[c]);      // ground contact: (Dc_a s_a)_v . sv_i
        }
        const double d_lo = d_a - 2.0 * d_b;
        const double mu = (double)(unsigned)((desc_m >> I) & 1ull);
        const double ml = (double)(unsigned)((anc_m >> I) & 1ull);
        Mrow[I] = (I == lane) ? mdiag : (mu * m_up + ml * m_lo);
        Drow[I] = (I == lane) ? ddiag : (mu * d_up + ml * d_lo);
        md_columns_dpp<NP, I + 1, CD>(cv, sw, sv, r1, r2w, r2v, desc_m, anc_m, lane, mdiag, ddiag, Mrow, Drow);
    }
}

// yc = Dc s for eval_MD<.., CD>: the contact damping blocks of the bodies (contact_body<2>, 21 numbers each) summed over the subtree
// through the LDS scan, times this node's screw.  Zero while no corner of the tree penetrates (fs.touched, wave-uniform).
template <int NP>
__device__ __forceinline__ void contact_damping_fold(const DevModel& M, double* __restrict__ sAcc, const int lane, const FrontState& fs,
                                                     double (&yc)[6]) {
    constexpr int CS = cstride(NP);
#pragma unroll
    for (int c = 0; c < 6; ++c) yc[c] = 0.0;
    if (!fs.touched) return;
    const bool act = fs.act;
    const int jj = act ? lane : 0;
    const double* cEnd = RMX_CONSTS(sAcc, M.n, NP) + (NCONST - 5) * CS;
    const double* cCon = cEnd + CS;
    const bool con = act && cCon[jj] != 0.0;
    const double sd[3] = {cCon[CS + jj], cCon[2 * CS + jj], cCon[3 * CS + jj]};
    const GroundC G = ground_of<NP>(M, sAcc, jj);
    const double s6[6] = {fs.sw[0], fs.sw[1], fs.sw[2], fs.sv[0], fs.sv[1], fs.sv[2]};
    double Fc[6], eVc, Dx[36];
    contact_body<2>(G, con, sd, fs.Rw, fs.pw, fs.phw, fs.phv, Fc, Dx, eVc);
    lds_subtree_sum<NP, 21>(M, sAcc, cEnd, lane, act, jj, &Dx[0]);
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        double a = 0.0;
#pragma unroll
        for (int c = 0; c < 6; ++c) a += Dx[r <= c ? sym21(r, c) : sym21(c, r)] * s6[c];
        yc[r] = a;
    }
}

// CD: models with ground contact.  yc = Dc s of this node, Dc = the contact damping blocks Dw = sum Gw' Y Gw of the bodies of its subtree
// (contact_damping_fold): D gains  s_a . (Dc_i s_i)  for i a descendant of a,  (Dc_a s_a) . s_i  for an ancestor,  s_a . Dc_a s_a  on the
// diagonal - the same places the Coriolis block Bc occupies, so yc joins y (Bc s) and its transpose-side twin r2.
template <int NP, bool CD = false>
__device__ __forceinline__ void eval_MD(const DevModel& M, const int lane, const FrontState& fs, double (&Mrow)[NP], double (&Drow)[NP],
                                        const double* yc = nullptr) {
    const bool act = fs.act;
    const int jj = act ? lane : 0;
    const double mS = fs.S[6];
    const double* mcS = &fs.S[7];
    const double* IbS = &fs.S[10];
    const double* TL = &fs.S[16];
    const double* hfS = &fs.S[25];
    double r1[6], ix[6], yD[6], r2w[3], t3[3], b3[3];
    sym3v(IbS, fs.sw, r1);                    // r1 = Ic s
    cross3(mcS, fs.sv, t3);
#pragma unroll
    for (int c = 0; c < 3; ++c) r1[c] += t3[c];
    cross3(mcS, fs.sw, t3);
#pragma unroll
    for (int c = 0; c < 3; ++c) r1[3 + c] = mS * fs.sv[c] - t3[c];
    sym3v(IbS, fs.xiw, ix);                   // ix = Ic xi
    cross3(mcS, fs.xiv, t3);
#pragma unroll
    for (int c = 0; c < 3; ++c) ix[c] += t3[c];
    cross3(mcS, fs.xiw, t3);
#pragma unroll
    for (int c = 0; c < 3; ++c) ix[3 + c] = mS * fs.xiv[c] - t3[c];
    mat3v(TL, fs.sw, yD);                     // Bc s = [TL sw ; 2 hf x sw]
    cross3(hfS, fs.sw, b3);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        yD[c] -= 2.0 * ix[c];
        yD[3 + c] = 2.0 * b3[c] - 2.0 * ix[3 + c];
    }
    cross3(hfS, fs.sv, b3);                   // r2w = (Bc' s)_w = TL' sw - 2 hf x sv
#pragma unroll
    for (int c = 0; c < 3; ++c) r2w[c] = TL[c] * fs.sw[0] + TL[3 + c] * fs.sw[1] + TL[6 + c] * fs.sw[2] - 2.0 * b3[c];
    double r2v[3] = {0.0, 0.0, 0.0};
    if constexpr (CD) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            yD[c] += yc[c];
            yD[3 + c] += yc[3 + c];
            r2w[c] += yc[c];          // Dc is symmetric: Dc' s = Dc s
            r2v[c] = yc[3 + c];
        }
    }
    double cv[24];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        cv[c] = act ? fs.sw[c] : 0.0;
        cv[3 + c] = act ? fs.sv[c] : 0.0;
        cv[6 + c] = act ? fs.xiw[c] : 0.0;
        cv[9 + c] = act ? fs.xiv[c] : 0.0;
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        cv[12 + c] = act ? r1[c] : 0.0;
        cv[18 + c] = act ? yD[c] : 0.0;
    }
    const unsigned long long anc_m = fs.anc_m, desc_m = fs.desc_m;
    const double sr1 = fs.sw[0] * r1[0] + fs.sw[1] * r1[1] + fs.sw[2] * r1[2] + fs.sv[0] * r1[3] + fs.sv[1] * r1[4] + fs.sv[2] * r1[5];
    const double syD = fs.sw[0] * yD[0] + fs.sw[1] * yD[1] + fs.sw[2] * yD[2] + fs.sv[0] * yD[3] + fs.sv[1] * yD[4] + fs.sv[2] * yD[5];
    const double mdiag = fs.dof ? sr1 : 1.0;
    const double ddiag = fs.dof ? (syD - fs.dd) : 0.0;
    if constexpr (NP <= 16) {     // one DPP row holds the tree: broadcasts fused into the FMAs (see eval_hess)
        dpp_settle(cv);
        md_columns_dpp<NP, 0, CD>(cv, fs.sw, fs.sv, r1, r2w, r2v, desc_m, anc_m, lane, mdiag, ddiag, Mrow, Drow);
        return;
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        double Ci[24];
#pragma unroll
        for (int c = 0; c < 24; ++c) Ci[c] = readlane_d(cv[c], i);
        const double m_lo = r1[0] * Ci[0] + r1[1] * Ci[1] + r1[2] * Ci[2] + r1[3] * Ci[3] + r1[4] * Ci[4] + r1[5] * Ci[5];
        const double m_up = fs.sw[0] * Ci[12] + fs.sw[1] * Ci[13] + fs.sw[2] * Ci[14] + fs.sv[0] * Ci[15] + fs.sv[1] * Ci[16] + fs.sv[2] * Ci[17];
        double d_lo = r2w[0] * Ci[0] + r2w[1] * Ci[1] + r2w[2] * Ci[2] -
                      2.0 * (r1[0] * Ci[6] + r1[1] * Ci[7] + r1[2] * Ci[8] + r1[3] * Ci[9] + r1[4] * Ci[10] + r1[5] * Ci[11]);
        if constexpr (CD) d_lo += r2v[0] * Ci[3] + r2v[1] * Ci[4] + r2v[2] * Ci[5];
        const double d_up = fs.sw[0] * Ci[18] + fs.sw[1] * Ci[19] + fs.sw[2] * Ci[20] + fs.sv[0] * Ci[21] + fs.sv[1] * Ci[22] + fs.sv[2] * Ci[23];
        const double mu = (double)(unsigned)((desc_m >> i) & 1ull);
        const double ml = (double)(unsigned)((anc_m >> i) & 1ull);
        Mrow[i] = (i == lane) ? mdiag : (mu * m_up + ml * m_lo);
        Drow[i] = (i == lane) ? ddiag : (mu * d_up + ml * d_lo);
    }
}

// M and D of a tree of up to 32 nodes on the fp64 matrix cores, for the adjoint's per-step history (k_adjoint_fwd<32>): the same four
// masked products eval_MD forms column by column with v_readlane broadcasts (24 scalar pairs + ~40 FMAs per column, 2.8 k issue slots
// per step and 0.2 - 1.1 KB of scratch in the 32-lane forward kernels),
//     M(a,i) = [a anc i] s_a . r1_i + [a desc i] r1_a . s_i                                   r1 = Ic s
//     D(a,i) = [a anc i] s_a . yD_i + [a desc i] (r2w_a . sw_i - 2 r1_a . xi_i)               yD = Bc s - 2 Ic xi, r2w = (Bc' s)_w
// as 2 x 2 tiles of v_mfma_f64_16x16x4_f64 with the operands staged in LDS in [k][node] order (eval_hess's scheme: a row of the
// staging area serves as A operand - lane (j, g) -> A[row 16 t + j][k 4 kk + g] - and as B operand alike), 27 MFMAs.  The results are
// stored to the history straight from the MFMA layout (lane (j, g), element r: row 16 mb + 4 r + g, column 16 nb + j), column-major
// [i * n + a] as the backward sweep reads them; nothing returns to row-per-lane.  Ends with sAcc handed back to the front.
__device__ __forceinline__ void eval_MD_mfma32_store(const DevModel& M, const int lane, const FrontState& fs, double* __restrict__ sAcc,
                                                     double* __restrict__ Mk, double* __restrict__ Dk) {
    constexpr int NP = 32, CS = cstride(NP), ST = HM_OP_STRIDE;
    constexpr int R_S = 0, R_R1 = 8, R_YD = 16, R_R2 = 24, R_M2 = 32, R_XI = 40, R_MD = 48, R_DD = 49;
    static_assert(R_DD + 1 <= HM_ROWS, "operand rows");
    typedef double v4d __attribute__((ext_vector_type(4)));
    const int n = M.n;
    const double mS = fs.S[6];
    const double* mcS = &fs.S[7];
    const double* IbS = &fs.S[10];
    const double* TL = &fs.S[16];
    const double* hfS = &fs.S[25];
    double r1[6], ix[6], yD[6], r2w[3], t3[3], b3[3];
    sym3v(IbS, fs.sw, r1);                    // r1 = Ic s
    cross3(mcS, fs.sv, t3);
#pragma unroll
    for (int c = 0; c < 3; ++c) r1[c] += t3[c];
    cross3(mcS, fs.sw, t3);
#pragma unroll
    for (int c = 0; c < 3; ++c) r1[3 + c] = mS * fs.sv[c] - t3[c];
    sym3v(IbS, fs.xiw, ix);                   // ix = Ic xi
    cross3(mcS, fs.xiv, t3);
#pragma unroll
    for (int c = 0; c < 3; ++c) ix[c] += t3[c];
    cross3(mcS, fs.xiw, t3);
#pragma unroll
    for (int c = 0; c < 3; ++c) ix[3 + c] = mS * fs.xiv[c] - t3[c];
    mat3v(TL, fs.sw, yD);                     // Bc s = [TL sw ; 2 hf x sw]
    cross3(hfS, fs.sw, b3);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        yD[c] -= 2.0 * ix[c];
        yD[3 + c] = 2.0 * b3[c] - 2.0 * ix[3 + c];
    }
    cross3(hfS, fs.sv, b3);                   // r2w = (Bc' s)_w = TL' sw - 2 hf x sv
#pragma unroll
    for (int c = 0; c < 3; ++c) r2w[c] = TL[c] * fs.sw[0] + TL[3 + c] * fs.sw[1] + TL[6 + c] * fs.sw[2] - 2.0 * b3[c];
    const double sr1 = fs.sw[0] * r1[0] + fs.sw[1] * r1[1] + fs.sw[2] * r1[2] + fs.sv[0] * r1[3] + fs.sv[1] * r1[4] + fs.sv[2] * r1[5];
    const double syD = fs.sw[0] * yD[0] + fs.sw[1] * yD[1] + fs.sw[2] * yD[2] + fs.sv[0] * yD[3] + fs.sv[1] * yD[4] + fs.sv[2] * yD[5];
    double* sOp = sAcc;
    if (lane < NP) {      // idle node slots n .. NP-1 carry s = 0, hence all-zero vectors by arithmetic
        double* o = sOp + lane;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            o[(R_S + c) * ST] = fs.sw[c];
            o[(R_S + 3 + c) * ST] = fs.sv[c];
            o[(R_R2 + c) * ST] = r2w[c];
            o[(R_R2 + 3 + c) * ST] = 0.0;
            o[(R_XI + c) * ST] = fs.xiw[c];
            o[(R_XI + 3 + c) * ST] = fs.xiv[c];
        }
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            o[(R_R1 + c) * ST] = r1[c];
            o[(R_YD + c) * ST] = yD[c];
            o[(R_M2 + c) * ST] = -2.0 * r1[c];
        }
#pragma unroll
        for (int b = 0; b < 6; ++b) {         // K = 6 -> 8: two zero rows per operand
            o[(8 * b + 6) * ST] = 0.0;
            o[(8 * b + 7) * ST] = 0.0;
        }
        o[R_MD * ST] = fs.dof ? sr1 : 1.0;
        o[R_DD * ST] = fs.dof ? (syD - fs.dd) : 0.0;
    }
    RMX_SYNC();
    const int g = lane >> 4, j = lane & 15;
    v4d mu[2][2], ml[2][2], du[2][2], dl[2][2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) mu[mb][nb] = ml[mb][nb] = du[mb][nb] = dl[mb][nb] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        double as[2], ar1[2], ar2[2], am2[2], bs[2], br1[2], byd[2], bxi[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int col = 16 * t + j, k = 4 * kk + g;
            as[t] = sOp[(R_S + k) * ST + col];
            ar1[t] = sOp[(R_R1 + k) * ST + col];
            ar2[t] = sOp[(R_R2 + k) * ST + col];
            am2[t] = sOp[(R_M2 + k) * ST + col];
            byd[t] = sOp[(R_YD + k) * ST + col];
            bxi[t] = sOp[(R_XI + k) * ST + col];
            bs[t] = as[t];
            br1[t] = ar1[t];
        }
        // depth-first numbering: an ancestor has the smaller index - the UP tile (rows 16..31, columns 0..15) and the LO tile (rows
        // 0..15, columns 16..31) are empty
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                if (nb >= mb) {
                    mu[mb][nb] = __builtin_amdgcn_mfma_f64_16x16x4f64(as[mb], br1[nb], mu[mb][nb], 0, 0, 0);
                    du[mb][nb] = __builtin_amdgcn_mfma_f64_16x16x4f64(as[mb], byd[nb], du[mb][nb], 0, 0, 0);
                }
                if (nb <= mb) {
                    ml[mb][nb] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar1[mb], bs[nb], ml[mb][nb], 0, 0, 0);
                    dl[mb][nb] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar2[mb], bs[nb], dl[mb][nb], 0, 0, 0);
                    dl[mb][nb] = __builtin_amdgcn_mfma_f64_16x16x4f64(am2[mb], bxi[nb], dl[mb][nb], 0, 0, 0);
                }
            }
    }
    const double* cRel = RMX_CONSTS(sAcc, M.n, NP) + (36 + 6 + 4 + 8 + 1) * CS;   // relation bit masks of the nodes (as doubles)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int col = 16 * nb + j;
        const unsigned am = (unsigned)((unsigned long long)__double_as_longlong(cRel[col]) >> g);        // bit a: a is a strict ancestor of col
        const unsigned dm = (unsigned)((unsigned long long)__double_as_longlong(cRel[CS + col]) >> g);   // ... a strict descendant
        const double md = sOp[R_MD * ST + col], dd = sOp[R_DD * ST + col];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * mb + 4 * r + g, sh = 16 * mb + 4 * r;
                const double wa = (double)((am >> sh) & 1u), wd = (double)((dm >> sh) & 1u);
                double mv = 0.0, dv = 0.0;
                if (nb >= mb) { mv = wa * mu[mb][nb][r]; dv = wa * du[mb][nb][r]; }
                if (nb <= mb) { mv += wd * ml[mb][nb][r]; dv += wd * dl[mb][nb][r]; }
                if (row == col) { mv = md; dv = dd; }
                if (row < n && col < n) {
                    Mk[(size_t)col * n + row] = mv;
                    Dk[(size_t)col * n + row] = dv;
                }
            }
    }
    RMX_SYNC();                 // sAcc goes back to the front, whose subtree scan relies on a zero row n
    if (lane < ACC_STRIDE) sAcc[M.n * ACC_STRIDE + lane] = 0.0;
    RMX_SYNC();
}

// Trees of 33..64 nodes: a 64 x 32 half of H (all rows x the columns i = 2 t + W) on the fp64 matrix cores, from operands staged
// in LDS ([k][node], stride H64_OP_STRIDE):
//     H(a,i) = [a strict ancestor of i] RU_a.CU_i + [a strict descendant of i] RL_a.CL_i ,  Hdiag on the diagonal,
// RU = s (6), CU = y - z (6), RL = (r1, -r2w, -r3w) (12), CL = (m1, m2w, sw) (12), as for n <= 32 in eval_hess.  4 x 2 tiles of
// 16 x 16; in depth-first numbering an ancestor has the smaller index, so 2 of the 8 UP tiles and 2 of the 8 LO tiles are empty:
// 6 x 2 + 6 x 3 = 30 v_mfma_f64_16x16x4_f64 per half; eval_hess computes both halves, one after the other.
constexpr int H64_R_RU = 0, H64_R_RL = 6, H64_R_CU = 18, H64_R_CL = 24, H64_R_HD = 36;
// The 64 x 32 half W of H on the matrix cores (columns i = 2 t + W), masked: hv[mb][nb][r] = H(16 mb + 4 r + g, 32 nb + 2 j + W)
// for g = lane >> 4, j = lane & 15 (the C/D layout of the f64 MFMA).  cRel: the relation-mask rows of the per-node constants.
template <int NP, int W>
__device__ __forceinline__ void hess64_tiles(const int lane, const double* __restrict__ sOp, const double* __restrict__ cRel,
                                              double (&hv)[4][2][4]) {
    static_assert(NP == 64, "64-lane trees");
    constexpr int ST = H64_OP_STRIDE;
    typedef double v4d __attribute__((ext_vector_type(4)));
    const int g = lane >> 4, j = lane & 15;
    constexpr int CSW = cstride(NP);
    v4d up[4][2], lw[4][2];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            up[mb][nb] = v4d{0.0, 0.0, 0.0, 0.0};
            lw[mb][nb] = v4d{0.0, 0.0, 0.0, 0.0};
        }
    // tile (mb, nb): rows 16 mb .. +15, column nodes 32 nb + W .. 32 nb + 30 + W.  UP needs a row below some column
    // (16 mb < 32 nb + 31), LO a row above some column (16 mb + 15 > 32 nb)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {       // UP: K = 6, padded to 8 by zero operands on the lanes with k = 6, 7
        double a[4], b[2];
        const bool kon = 4 * kk + g < 6;
        const int kr = kon ? 4 * kk + g : 0;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) a[mb] = kon ? sOp[(H64_R_RU + kr) * ST + 16 * mb + j] : 0.0;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) b[nb] = kon ? sOp[(H64_R_CU + kr) * ST + 32 * nb + 2 * j + W] : 0.0;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
                if (16 * mb < 32 * nb + 31) up[mb][nb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mb], b[nb], up[mb][nb], 0, 0, 0);
    }
#pragma unroll
    for (int kk = 0; kk < 3; ++kk) {       // LO: K = 12
        double a[4], b[2];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) a[mb] = sOp[(H64_R_RL + 4 * kk + g) * ST + 16 * mb + j];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) b[nb] = sOp[(H64_R_CL + 4 * kk + g) * ST + 32 * nb + 2 * j + W];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
                if (16 * mb + 15 > 32 * nb) lw[mb][nb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mb], b[nb], lw[mb][nb], 0, 0, 0);
    }
    // relation bits of this lane's two column nodes i = 32 nb + 2 j + W: bit a of its ancestor mask -> UP applies to row a,
    // of its descendant mask -> LO applies; rows a = 16 mb + 4 r + g: low word for mb < 2, high word otherwise
    unsigned amlo[2], amhi[2], dmlo[2], dmhi[2];
    double hd[2];
    int icol[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        icol[nb] = 32 * nb + 2 * j + W;
        const unsigned long long am = (unsigned long long)__double_as_longlong(cRel[icol[nb]]) >> g;
        const unsigned long long dm = (unsigned long long)__double_as_longlong(cRel[CSW + icol[nb]]) >> g;
        amlo[nb] = (unsigned)am;
        amhi[nb] = (unsigned)(am >> 32);
        dmlo[nb] = (unsigned)dm;
        dmhi[nb] = (unsigned)(dm >> 32);
        hd[nb] = sOp[H64_R_HD * ST + icol[nb]];
    }
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int sh = (16 * mb + 4 * r) & 31;
                const unsigned aw = mb < 2 ? amlo[nb] : amhi[nb], dw = mb < 2 ? dmlo[nb] : dmhi[nb];
                double v = 0.0;
                if (16 * mb < 32 * nb + 31) v = (double)((aw >> sh) & 1u) * up[mb][nb][r];
                if (16 * mb + 15 > 32 * nb) v += (double)((dw >> sh) & 1u) * lw[mb][nb][r];
                hv[mb][nb][r] = (16 * mb + 4 * r + g == icol[nb]) ? hd[nb] : v;
            }
}

// RMX_W2: the helper wave's own area of the workgroup's LDS, behind the per-node constants: an accumulation scratch for residual-only
// evaluations (row stride 7: the six sums of a node), the arguments of one evaluation (x, qdot, v per lane, eta) and its results
constexpr int W2_HELP_AS = 7;
constexpr int W2_HELP_ARGS = (MAXN + 1) * W2_HELP_AS + 1;
constexpr int W2_HELP_RES = W2_HELP_ARGS + 3 * 64 + 2;         // |g|^2, T, V
constexpr int W2_HELP_DOUBLES = W2_HELP_RES + 4;
// ... and of a workgroup of chains of <= 32 nodes (rmx_kernels.hip RMX_PART 6), whose helper runs the FULL evaluation (the chain's
// residual-only front sums by a register scan: other last bits) on a full-stride scratch
constexpr int W2C_HELP_ARGS = (32 + 1) * ACC_STRIDE + 1;
constexpr int W2C_HELP_RES = W2C_HELP_ARGS + 3 * 64 + 2;
constexpr int W2C_HELP_DOUBLES = W2C_HELP_RES + 4;
// RMX_W2: the command word of a two-wave workgroup (1: a Hessian stage and a guarded solve follow, 2: one residual-only evaluation,
// 0: the rollout is over)
__device__ __forceinline__ volatile int* w2_cmd() {
    __shared__ int cmd[2];
    return cmd;
}
// ... and one wave's column half of H (hess64_tiles<NP, W>) into the row-major matrix
template <int W>
__device__ __forceinline__ void w2_store_half(double* __restrict__ sAcc, const int lane, const double (&hv)[4][2][4]) {
    const int g = lane >> 4, j = lane & 15;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) sAcc[(16 * mb + 4 * r + g) * H64_STRIDE + 32 * nb + 2 * j + W] = hv[mb][nb][r];
}

// Hessian row of this node: Hrow[i] = H(row of this node, column of node i); rows/columns of idle node slots are the identity.
// Returns H(lane,lane).  ZERO_IDLE = false (n <= 32 MFMA path only): lanes 32..63 are left with mirrored rows instead of zeros.
// Column I (and on) of H for trees of up to 16 nodes, see eval_hess: H(a, i) = [i strict descendant of a] s_a . cu_i +
// [i strict ancestor of a] (r1_a . m1_i - r2w_a . m2w_i - r3w_a . sw_i  - contact terms), Hdiag on the diagonal.
template <int NP, bool CT, int I, int NCV>
__device__ __forceinline__ void hess_columns_dpp(const double (&cv)[NCV], const double (&sw)[3], const double (&sv)[3], const double (&r1t)[3],
                                                 const double (&r1f)[3], const double (&r2w)[3], const double (&r3w)[3],
                                                 const double (&cxr2)[6], const double (&cxr3)[6], const unsigned long long desc_m,
                                                 const unsigned long long anc_m, const int lane, const double Hdiag, double (&Hrow)[NP]) {
    if constexpr (I < NP) {
        // three accumulators taken in turn: two consecutive FMAs into the same register cost a wait state each
        double up = 0.0, lo = 0.0, lo2 = 0.0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            fmadd_rowbcast<I>(up, cv[c], sw[c]);
            fmadd_rowbcast<I>(lo, cv[6 + c], r1t[c]);
            fmsub_rowbcast<I>(lo2, cv[12 + c], r2w[c]);
            fmadd_rowbcast<I>(up, cv[3 + c], sv[c]);
            fmadd_rowbcast<I>(lo, cv[9 + c], r1f[c]);
            fmsub_rowbcast<I>(lo2, cv[15 + c], r3w[c]);
            if constexpr (CT) {
                fmsub_rowbcast<I>(lo, cv[18 + c], cxr2[3 + c]);
                fmsub_rowbcast<I>(lo2, cv[21 + c], cxr3[3 + c]);
            }
        }
        lo += lo2;
        const double mu = (double)(unsigned)((desc_m >> I) & 1ull);   // column node I is a strict descendant of this row's node
        const double ml = (double)(unsigned)((anc_m >> I) & 1ull);    // column node I is a strict ancestor
        const double hv = mu * up + ml * lo;
        Hrow[I] = (I == lane) ? Hdiag : hv;
        hess_columns_dpp<NP, CT, I + 1>(cv, sw, sv, r1t, r1f, r2w, r3w, cxr2, cxr3, desc_m, anc_m, lane, Hdiag, Hrow);
    }
}

// PRIMSEL (n <= 32, the pair front of rmx_pair32.h: lanes 0..31 and 32..63 hold the state of two different points, node = lane & 31):
// the half-wave `prim` stages its operands and right-hand side; everything after the staging is the same code.
template <int NP, bool TIMED = false, bool CT = false, bool ZERO_IDLE = true, bool PRIMSEL = false>
__device__ __forceinline__ double eval_hess(const DevModel& M, const int lane, const FrontState& fs, double (&Hrow)[NP],
                                          unsigned long long* stamps = nullptr, double* __restrict__ sAcc = nullptr, const double g_stage = 0.0,
                                          const int prim = 0) {
    static_assert(!PRIMSEL || (NP == 32 && HESS_MFMA && !CT), "PRIMSEL: the plain matrix-core Hessian of n <= 32");
    (void)prim;
    unsigned long long last_ = TIMED ? __builtin_amdgcn_s_memtime() : 0ull;
    constexpr int CS = cstride(NP);
    const double eta = fs.eta, e2 = eta * eta;
    const bool act = fs.act, dof = fs.dof;
    const int jj = act ? lane : 0;
    const double (&sw)[3] = fs.sw;
    const double (&sv)[3] = fs.sv;
    const double (&phw)[3] = fs.phw;
    const double (&phv)[3] = fs.phv;
    const double (&xiw)[3] = fs.xiw;
    const double (&xiv)[3] = fs.xiv;
    // ground contact: K/D blocks of this body at the state the front left behind, summed over the subtree (72 numbers through
    // the LDS scan in chunks) and folded at once into the three vectors the Hessian needs:
    //   cxy = Dxc m2 + eta^2 Kxc s ,  cxr2 = Dxc' s ,  cxr3 = eta^2 Kxc' s      (m2 = eta s + eta^2 xi)
    // skipped (wave-uniform) while no corner of the tree penetrates.
    double cxy[6], cxr2[6], cxr3[6];
    if constexpr (CT) {
#pragma unroll
        for (int c = 0; c < 6; ++c) cxy[c] = cxr2[c] = cxr3[c] = 0.0;
        if (fs.touched) {
            const double* cEnd = RMX_CONSTS(sAcc, M.n, NP) + (NCONST - 5) * CS;
            const double* cCon = cEnd + CS;
            const bool con = act && cCon[jj] != 0.0;
            const double sd[3] = {cCon[CS + jj], cCon[2 * CS + jj], cCon[3 * CS + jj]};
            const GroundC G = ground_of<NP>(M, sAcc, jj);
            const double s6[6] = {sw[0], sw[1], sw[2], sv[0], sv[1], sv[2]};
            double m26[6];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                m26[c] = eta * sw[c] + e2 * xiw[c];
                m26[3 + c] = eta * sv[c] + e2 * xiv[c];
            }
            double Fc[6], eVc;
            {   // stiffness block: 36 numbers = two passes of the 28-wide scan
                double Kx[36];
                contact_body<1>(G, con, sd, fs.Rw, fs.pw, phw, phv, Fc, Kx, eVc);
                lds_subtree_sum<NP, 28>(M, sAcc, cEnd, lane, act, jj, &Kx[0]);
                lds_subtree_sum<NP, 8>(M, sAcc, cEnd, lane, act, jj, &Kx[28]);
#pragma unroll
                for (int r = 0; r < 6; ++r) {
                    double ay = 0.0, a3 = 0.0;
#pragma unroll
                    for (int c = 0; c < 6; ++c) {
                        ay += Kx[6 * r + c] * s6[c];
                        a3 += Kx[6 * c + r] * s6[c];
                    }
                    cxy[r] = e2 * ay;
                    cxr3[r] = e2 * a3;
                }
            }
            RMX_STAMP(12)
            {   // damping block, symmetric: 21 numbers, one pass
                double Dx[36];
                contact_body<2>(G, con, sd, fs.Rw, fs.pw, phw, phv, Fc, Dx, eVc);
                lds_subtree_sum<NP, 21>(M, sAcc, cEnd, lane, act, jj, &Dx[0]);
#pragma unroll
                for (int r = 0; r < 6; ++r) {
                    double ay = 0.0, a2 = 0.0;
#pragma unroll
                    for (int c = 0; c < 6; ++c) {
                        const double drc = Dx[r <= c ? sym21(r, c) : sym21(c, r)];
                        ay += drc * m26[c];
                        a2 += drc * s6[c];
                    }
                    cxy[r] += ay;
                    cxr2[r] = a2;
                }
            }
            RMX_STAMP(13)
        }
    }
    const double (&bw)[3] = fs.bw;
    const double (&bv)[3] = fs.bv;
    const double* Wt = &fs.S[0];
    const double* Wf = &fs.S[3];
    const double mS = fs.S[6];
    const double* mcS = &fs.S[7];
    const double* IbS = &fs.S[10];
    const double* TL = &fs.S[16];
    const double* hfS = &fs.S[25];
    const double gv[3] = {M.grav[0], M.grav[1], M.grav[2]};
    double t3[3], a3[3], b3[3];
    // zeta = ad(beta) s + eta^2 ad(phi) xi ; m1 = s + 2 eta xi + zeta ; m2w = eta sw + eta^2 xiw
    double zw[3], zv[3], m1w[3], m1v[3], m2w[3];
    cross3(bw, sw, zw);
    cross3(bv, sw, zv);
    cross3(bw, sv, t3);
#pragma unroll
    for (int c = 0; c < 3; ++c) zv[c] += t3[c];
    cross3(phw, xiw, a3);
    cross3(phv, xiw, b3);
    cross3(phw, xiv, t3);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        zw[c] += e2 * a3[c];
        zv[c] += e2 * (b3[c] + t3[c]);
        m1w[c] = sw[c] + 2.0 * eta * xiw[c] + zw[c];
        m1v[c] = sv[c] + 2.0 * eta * xiv[c] + zv[c];
        m2w[c] = eta * sw[c] + e2 * xiw[c];
    }
    // y = Ic m1 - Bc m2 - eta^2 Kc s
    double yt[3], yf[3];
    sym3v(IbS, m1w, yt);
    cross3(mcS, m1v, t3);
#pragma unroll
    for (int c = 0; c < 3; ++c) yt[c] += t3[c];
    cross3(mcS, m1w, t3);
#pragma unroll
    for (int c = 0; c < 3; ++c) yf[c] = mS * m1v[c] - t3[c];
    mat3v(TL, m2w, a3);                 // Bc m2 : top = TL m2w, bottom = 2 hf x m2w
    cross3(hfS, m2w, b3);
    double gxs[3], kt[3];
    cross3(gv, sw, gxs);                // Kc s : top = mc x (g x sw), bottom = m g x sw
    cross3(mcS, gxs, kt);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        yt[c] -= a3[c] + e2 * kt[c];
        yf[c] -= 2.0 * b3[c] + e2 * mS * gxs[c];
        if (CT) {
            yt[c] -= cxy[c];
            yf[c] -= cxy[3 + c];
        }
    }
    // z = ad(s)' W
    double zt[3], zf[3];
    cross3(sw, Wt, a3);
    cross3(sv, Wf, b3);
    cross3(sw, Wf, zf);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        zt[c] = -a3[c] - b3[c];
        zf[c] = -zf[c];
    }
    // diagonal: s.y - eta Dr - eta^2 Kr   (Joint.computeForce Joint.m:470-482)
    const double Hdiag = dof ? (dot3(sw, yt) + dot3(sv, yf) + eta * fs.dd + e2 * fs.kd) : 1.0;
    // row-side vectors: r1 = Ic s, r2w = TL' sw - 2 hf x sv, r3w = eta^2 (g x (mc x sw) - m g x sv)
    double r1t[3], r1f[3], r2w[3], r3w[3];
    sym3v(IbS, sw, r1t);
    cross3(mcS, sv, t3);
#pragma unroll
    for (int c = 0; c < 3; ++c) r1t[c] += t3[c];
    cross3(mcS, sw, t3);
#pragma unroll
    for (int c = 0; c < 3; ++c) r1f[c] = mS * sv[c] - t3[c];
    cross3(hfS, sv, b3);
#pragma unroll
    for (int c = 0; c < 3; ++c) r2w[c] = TL[c] * sw[0] + TL[3 + c] * sw[1] + TL[6 + c] * sw[2] - 2.0 * b3[c];
    cross3(gv, t3, a3);     // g x (mc x sw)
    cross3(gv, sv, b3);
#pragma unroll
    for (int c = 0; c < 3; ++c) r3w[c] = e2 * (a3[c] - mS * b3[c]);
    if (CT) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            r2w[c] += cxr2[c];
            r3w[c] += cxr3[c];
        }
    }
    RMX_STAMP(9)
    // column-side vectors of this node (zero on idle lanes).  NP != 32: column i is broadcast out of lane i with v_readlane
    // into scalar registers, which the FMAs consume directly.  NP == 32: see below.
    constexpr int NCV = CT ? NCOLX : NCOL;
    // idle node slots have s = 0, hence all-zero column vectors by arithmetic; the explicit selects are kept only where the
    // vectors of lanes >= NP could be read (the v_readlane loops), not for the n <= 32 MFMA path, which stages lanes < NP only
    const bool keep = (NP == 32 && HESS_MFMA) ? true : act;
    double cv[NCV];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        if (CT) {
            cv[18 + c] = keep ? eta * sv[c] + e2 * xiv[c] : 0.0;   // m2v
            cv[21 + c] = keep ? sv[c] : 0.0;
        }
        cv[c] = keep ? yt[c] - zt[c] : 0.0;
        cv[3 + c] = keep ? yf[c] - zf[c] : 0.0;
        cv[6 + c] = keep ? m1w[c] : 0.0;
        cv[9 + c] = keep ? m1v[c] : 0.0;
        cv[12 + c] = keep ? m2w[c] : 0.0;
        cv[15 + c] = keep ? sw[c] : 0.0;
    }
    RMX_STAMP(10)
    const unsigned long long anc_m = fs.anc_m, desc_m = fs.desc_m;
    if constexpr (NP == 32 && HESS_MFMA) {
        // H as two small matrix products on the fp64 matrix cores.  With the row-side vectors RU_a = (s_a) [6], RL_a = (r1_a,
        // -r2w_a, -r3w_a [, -contact]) [12|18] and the column-side vectors CU_i = (y_i - z_i) [6], CL_i = (m1_i, m2w_i, sw_i
        // [, m2v_i, sv_i]) [12|18]:   H(a,i) = [a strict ancestor of i] RU_a.CU_i + [a strict descendant of i] RL_a.CL_i, Hdiag
        // on the diagonal.  UP = RU CU' (K = 6 -> 8) and LO = RL CL' (K = 12|18 -> 12|20) are 32x32 products =
        // 2x2 tiles of v_mfma_f64_16x16x4_f64, 2 + 3|5 instructions per tile.  Operands go through LDS in [k][node] order (A:
        // lane l supplies A[row l&15][k l>>4], B: B[k l>>4][col l&15]); each lane then owns columns c = 16 nb + (l&15) and rows
        // a = 16 mb + (l>>4) + 4 r of the results (C/D layout of the f64 MFMA), masks them with the relation bits of ITS TWO
        // COLUMN nodes, and the matrix returns to row-per-lane through LDS for the solve.  All 64 lanes work; n <= 32.
        constexpr int KL = CT ? 20 : 12;                 // padded K of the LO product
        constexpr int R_RU = 0, R_RL = 8, R_CU = 8 + KL, R_CL = 16 + KL, R_HD = 16 + 2 * KL;   // operand rows in LDS
        static_assert(R_HD + 1 <= HM_ROWS, "operand rows");
        double* sOp = sAcc;
        bool stage = lane < NP;          // the lanes that hold the nodes' state ...
        int nd = lane;                   // ... and their node
        if constexpr (PRIMSEL) {
            stage = (lane >> 5) == prim;
            nd = lane & 31;
        }
        if (stage) {
            double* o = sOp + nd;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                o[(R_RU + c) * HM_OP_STRIDE] = sw[c];
                o[(R_RU + 3 + c) * HM_OP_STRIDE] = sv[c];
                o[(R_RL + c) * HM_OP_STRIDE] = r1t[c];
                o[(R_RL + 3 + c) * HM_OP_STRIDE] = r1f[c];
                o[(R_RL + 6 + c) * HM_OP_STRIDE] = -r2w[c];
                o[(R_RL + 9 + c) * HM_OP_STRIDE] = -r3w[c];
                if (CT) {
                    o[(R_RL + 12 + c) * HM_OP_STRIDE] = -cxr2[3 + c];
                    o[(R_RL + 15 + c) * HM_OP_STRIDE] = -cxr3[3 + c];
                }
            }
            o[(R_RU + 6) * HM_OP_STRIDE] = 0.0;
            o[(R_RU + 7) * HM_OP_STRIDE] = 0.0;
            o[(R_CU + 6) * HM_OP_STRIDE] = 0.0;
            o[(R_CU + 7) * HM_OP_STRIDE] = 0.0;
            if (CT) {
                o[(R_RL + 18) * HM_OP_STRIDE] = 0.0;
                o[(R_RL + 19) * HM_OP_STRIDE] = 0.0;
                o[(R_CL + 18) * HM_OP_STRIDE] = 0.0;
                o[(R_CL + 19) * HM_OP_STRIDE] = 0.0;
            }
#pragma unroll
            for (int c = 0; c < 6; ++c) o[(R_CU + c) * HM_OP_STRIDE] = cv[c];
#pragma unroll
            for (int c = 6; c < NCV; ++c) o[(R_CL + c - 6) * HM_OP_STRIDE] = cv[c];
            o[R_HD * HM_OP_STRIDE] = Hdiag;
        }
        RMX_SYNC();
        typedef double v4d __attribute__((ext_vector_type(4)));
        const int g = lane >> 4, j = lane & 15;
        const double* cRel = RMX_CONSTS(sAcc, M.n, NP) + (36 + 6 + 4 + 8 + 1) * CS;   // relation bit masks of the nodes (as doubles)
        v4d up[2][2], lw[2][2];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                up[mb][nb] = v4d{0.0, 0.0, 0.0, 0.0};
                lw[mb][nb] = v4d{0.0, 0.0, 0.0, 0.0};
            }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            double a[2], b[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                a[t] = sOp[(R_RU + 4 * kk + g) * HM_OP_STRIDE + 16 * t + j];
                b[t] = sOp[(R_CU + 4 * kk + g) * HM_OP_STRIDE + 16 * t + j];
            }
            // depth-first numbering: an ancestor has the smaller index, so the UP tile (rows 16..31, columns 0..15) and the
            // LO tile (rows 0..15, columns 16..31) are masked out entirely and are not computed
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = mb; nb < 2; ++nb) up[mb][nb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mb], b[nb], up[mb][nb], 0, 0, 0);
        }
#pragma unroll
        for (int kk = 0; kk < KL / 4; ++kk) {
            double a[2], b[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                a[t] = sOp[(R_RL + 4 * kk + g) * HM_OP_STRIDE + 16 * t + j];
                b[t] = sOp[(R_CL + 4 * kk + g) * HM_OP_STRIDE + 16 * t + j];
            }
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb <= mb; ++nb) lw[mb][nb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mb], b[nb], lw[mb][nb], 0, 0, 0);
        }
        // relation bits of this lane's two column nodes c = 16 nb + j: bit a of the ancestor mask -> UP applies, of the
        // descendant mask -> LO applies; shifted by g so that the row a = 16 mb + 4 r + g needs a constant shift
        unsigned am[2], dm[2];        // rows 16 mb + 4 r + g <= 31: the low words of the shifted masks are enough
        double hd[2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            am[nb] = (unsigned)((unsigned long long)__double_as_longlong(cRel[16 * nb + j]) >> g);
            dm[nb] = (unsigned)((unsigned long long)__double_as_longlong(cRel[CS + 16 * nb + j]) >> g);
            hd[nb] = sOp[R_HD * HM_OP_STRIDE + 16 * nb + j];
        }
        double hv[2][2][4];
        if (M.is_chain) {
            // a chain in depth-first order: a is an ancestor of i iff a < i.  The off-diagonal tiles need no mask at all and
            // the diagonal tiles one comparison of the row 4 r + g with the column j (same values: the masks are 0 / 1)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (nb > mb) hv[mb][nb][r] = up[mb][nb][r];
                        else if (nb < mb) hv[mb][nb][r] = lw[mb][nb][r];
                        else hv[mb][nb][r] = (4 * r + g < j) ? up[mb][nb][r] : ((4 * r + g == j) ? hd[nb] : lw[mb][nb][r]);
                    }
        } else {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int sh = 16 * mb + 4 * r;
                    double v = 0.0;
                    if (nb >= mb) v = (double)((am[nb] >> sh) & 1u) * up[mb][nb][r];
                    if (nb <= mb) v += (double)((dm[nb] >> sh) & 1u) * lw[mb][nb][r];
                    hv[mb][nb][r] = (mb == nb && 4 * r + g == j) ? hd[nb] : v;
                }
        }
        RMX_SYNC();             // every lane is done with the operands: the same LDS now takes H, row-major
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) sOp[(16 * mb + 4 * r + g) * HM_H_STRIDE + 16 * nb + j] = hv[mb][nb][r];
        // the right-hand side of the solve travels with its row (column 32 of the staging rows is spare)
        if constexpr (!ZERO_IDLE && LU_SPLIT32)
            if (stage) sOp[nd * HM_H_STRIDE + 32] = -g_stage;
        RMX_SYNC();
        // guarded diagonal solve of n <= 32 (lu_solve_neg_diag32): it reads H out of the staging area in its own layout and
        // hands sAcc back to the front itself
        if constexpr (!ZERO_IDLE && LU_SPLIT32) return Hdiag;
        {
            typedef double v2d __attribute__((ext_vector_type(2)));
            const v2d* hr = reinterpret_cast<const v2d*>(sOp + (lane & 31) * HM_H_STRIDE);   // rows are 16-byte aligned
            const bool lo_half = lane < 32;
#pragma unroll
            for (int c = 0; c < NP / 2; ++c) {
                const v2d t = hr[c];
                // ZERO_IDLE: lanes >= 32 hold all-zero rows (the pivot search of lu_solve_neg looks at every lane).  The guarded
                // diagonal solve never reads them and ignores their guard bits, so there they keep the mirrored row.
                Hrow[2 * c] = (ZERO_IDLE && !lo_half) ? 0.0 : t[0];
                Hrow[2 * c + 1] = (ZERO_IDLE && !lo_half) ? 0.0 : t[1];
            }
        }
        RMX_SYNC();             // sAcc goes back to the front, whose subtree scan relies on a zero row n
        if (lane < ACC_STRIDE) sAcc[M.n * ACC_STRIDE + lane] = 0.0;
        RMX_SYNC();
    } else if constexpr (NP == 32) {
        // Trees of at most 32 nodes leave lanes 32..63 idle: they mirror the row-side state of lanes 0..31 and take columns
        // 16..31 while lanes 0..31 take columns 0..15, so the column loop runs 16 times instead of 32.  The column vectors go
        // through LDS (each half-wave reads ONE address per column: two-address broadcast, no bank conflict); the upper
        // half's 16 results come back with v_permlane32_swap.  sAcc is free here (the front is done with it).
        if (lane < NP) {
#pragma unroll
            for (int c = 0; c < NCV; ++c) sAcc[lane * NCV + c] = cv[c];
        }
        RMX_SYNC();
        const bool hi = lane >= 32;
        double rsw[3], rsv[3], q1t[3], q1f[3], q2w[3], q3w[3], x2v[3], x3v[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            rsw[c] = dup_lo(sw[c]);
            rsv[c] = dup_lo(sv[c]);
            q1t[c] = dup_lo(r1t[c]);
            q1f[c] = dup_lo(r1f[c]);
            q2w[c] = dup_lo(r2w[c]);
            q3w[c] = dup_lo(r3w[c]);
            if (CT) {
                x2v[c] = dup_lo(cxr2[3 + c]);
                x3v[c] = dup_lo(cxr3[3 + c]);
            }
        }
        const double hd = dup_lo(Hdiag);
        // relation masks of the mirrored row, shifted so that bit i is this half's column i
        const unsigned long long am = __double_as_longlong(dup_lo(__longlong_as_double((long long)anc_m))) >> (hi ? 16 : 0);
        const unsigned long long dm = __double_as_longlong(dup_lo(__longlong_as_double((long long)desc_m))) >> (hi ? 16 : 0);
        const int drow = (lane & 31) - (hi ? 16 : 0);                 // the row's own column, counted inside this half
        const double* cb = sAcc + (hi ? 16 * NCV : 0);
        double Hh[16], Cn[NCV];
#pragma unroll
        for (int c = 0; c < NCV; ++c) Cn[c] = cb[c];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            double Ci[NCV];
#pragma unroll
            for (int c = 0; c < NCV; ++c) Ci[c] = Cn[c];
            if (i + 1 < 16) {                  // two columns in flight
#pragma unroll
                for (int c = 0; c < NCV; ++c) Cn[c] = cb[(i + 1) * NCV + c];
            }
            const double up = rsw[0] * Ci[0] + rsw[1] * Ci[1] + rsw[2] * Ci[2] + rsv[0] * Ci[3] + rsv[1] * Ci[4] + rsv[2] * Ci[5];
            double lo = q1t[0] * Ci[6] + q1t[1] * Ci[7] + q1t[2] * Ci[8] + q1f[0] * Ci[9] + q1f[1] * Ci[10] + q1f[2] * Ci[11] -
                        (q2w[0] * Ci[12] + q2w[1] * Ci[13] + q2w[2] * Ci[14]) - (q3w[0] * Ci[15] + q3w[1] * Ci[16] + q3w[2] * Ci[17]);
            if (CT) lo -= x2v[0] * Ci[18] + x2v[1] * Ci[19] + x2v[2] * Ci[20] + x3v[0] * Ci[21] + x3v[1] * Ci[22] + x3v[2] * Ci[23];
            const double mu = (double)(unsigned)((dm >> i) & 1ull);
            const double ml = (double)(unsigned)((am >> i) & 1ull);
            const double hv = mu * up + ml * lo;
            Hh[i] = (i == drow) ? hd : hv;
            // pins column i's arithmetic between the loads of columns i+1 and i+2: otherwise the scheduler requests every
            // column first and spills hundreds of registers
            asm volatile("" : "+v"(Hh[i]) : : "memory");
        }
        // rows live in lanes 0..31: columns 16..31 come over from the upper half; lanes 32..63 go back to all-zero rows
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const double t = take_hi(Hh[i]);
            Hrow[i] = hi ? 0.0 : Hh[i];
            Hrow[16 + i] = hi ? 0.0 : t;
        }
        RMX_SYNC();             // sAcc goes back to the front, whose subtree scan relies on a zero row n
        if (lane < ACC_STRIDE) sAcc[M.n * ACC_STRIDE + lane] = 0.0;
        RMX_SYNC();
    } else if constexpr (NP == 64 && !CT && HESS_MFMA64) {
        // 33..64 nodes, one wavefront: the two matrix products of the two-wave kernel, both column halves by this wave.  Operands
        // in [k][node] order in the scratch, 2 x 30 v_mfma_f64_16x16x4_f64, then H row-major [64][H64_STRIDE] in the same scratch
        // (the layout lu_solve_neg_diag64 eliminates in; its right-hand side -g travels in column 64).
        constexpr int ST = H64_OP_STRIDE;
        double* sOp = sAcc;
        {
            double* o = sOp + lane;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                o[(H64_R_RU + c) * ST] = sw[c];
                o[(H64_R_RU + 3 + c) * ST] = sv[c];
                o[(H64_R_RL + c) * ST] = r1t[c];
                o[(H64_R_RL + 3 + c) * ST] = r1f[c];
                o[(H64_R_RL + 6 + c) * ST] = -r2w[c];
                o[(H64_R_RL + 9 + c) * ST] = -r3w[c];
            }
#pragma unroll
            for (int c = 0; c < 6; ++c) o[(H64_R_CU + c) * ST] = cv[c];
#pragma unroll
            for (int c = 6; c < 18; ++c) o[(H64_R_CL + c - 6) * ST] = cv[c];
            o[H64_R_HD * ST] = Hdiag;
        }
        if constexpr (!ZERO_IDLE) {
            if (M.tree_dmax > 0) {
                // a branching tree whose solve runs along the tree (tree_solve64): it needs H(i, ancestor) and H(ancestor, i) only -
                // 592 of configs[2]'s 4 096 entries - and forms them from these operands itself: no matrix products, no H in
                // LDS, and no second wavefront in the stage.  The right-hand side -g rides in the row behind the operands.
                sOp[H64_OP_ROWS * ST + lane] = -g_stage;
                RMX_SYNC();
                return Hdiag;
            }
        }
        const double* cRel = RMX_CONSTS(sAcc, M.n, NP) + (36 + 6 + 4 + 8 + 1) * CS;
        if constexpr (RMX_W2 && !ZERO_IDLE) {
            // two waves (w2_helper is the other one): this wave takes the even columns, the helper the odd ones
            if (lane == 0) *w2_cmd() = 1;
            RMX_WG_BAR();
            double h0[4][2][4];
            hess64_tiles<NP, 0>(lane, sOp, cRel, h0);
            RMX_WG_BAR();           // both waves are done with the operands: the same LDS now takes H
            w2_store_half<0>(sAcc, lane, h0);
            sAcc[lane * H64_STRIDE + 64] = -g_stage;
            RMX_WG_BAR();
            return Hdiag;
        }
        RMX_SYNC();
        double h0[4][2][4], h1[4][2][4];
        hess64_tiles<NP, 0>(lane, sOp, cRel, h0);
        __builtin_amdgcn_sched_barrier(0);      // one half's accumulators at a time
        hess64_tiles<NP, 1>(lane, sOp, cRel, h1);
        RMX_SYNC();             // every lane is done with the operands: the same LDS now takes H
        {
            typedef double v2d __attribute__((ext_vector_type(2)));
            const int g = lane >> 4, j = lane & 15;
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        *reinterpret_cast<v2d*>(sAcc + (16 * mb + 4 * r + g) * H64_STRIDE + 32 * nb + 2 * j) = v2d{h0[mb][nb][r], h1[mb][nb][r]};
            sAcc[lane * H64_STRIDE + 64] = -g_stage;
        }
        RMX_SYNC();
        if constexpr (!ZERO_IDLE) return Hdiag;         // the guarded solve takes H where it is (and hands the scratch back)
        {
            typedef double v2d __attribute__((ext_vector_type(2)));
            const v2d* hr = reinterpret_cast<const v2d*>(sAcc + lane * H64_STRIDE);
#pragma unroll
            for (int c = 0; c < NP / 2; ++c) {
                const v2d t = hr[c];
                Hrow[2 * c] = t[0];
                Hrow[2 * c + 1] = t[1];
            }
        }
        RMX_SYNC();             // sAcc goes back to the front, whose subtree scan relies on a zero row n
        if (lane < ACC_STRIDE) sAcc[M.n * ACC_STRIDE + lane] = 0.0;
        RMX_SYNC();
    } else if constexpr (NP <= 16) {
        // up to 16 nodes the whole tree sits in one 16-lane DPP row: the broadcast of column node i rides on the FMA
        // (v_fmac_f64_dpp row_newbcast, fmadd_rowbcast) - 18 instructions per column instead of 36 v_readlane + 18 FMA, and no
        // scalar registers: hoisted ahead of their FMAs, the broadcasts of 16 columns were 576 SGPRs, spilled to VGPR lanes and
        // back (1100 v_writelane / v_readlane pairs per call, two thirds of the adjoint forward kernel at 16 nodes).
        dpp_settle(cv);
        hess_columns_dpp<NP, CT, 0>(cv, sw, sv, r1t, r1f, r2w, r3w, cxr2, cxr3, desc_m, anc_m, lane, Hdiag, Hrow);
    } else {
#pragma unroll
        for (int t = 0; t < NP; ++t) {
            const int i = t;
            double Ci[NCV];
#pragma unroll
            for (int c = 0; c < NCV; ++c) Ci[c] = readlane_d(cv[c], i);
            const double up = sw[0] * Ci[0] + sw[1] * Ci[1] + sw[2] * Ci[2] + sv[0] * Ci[3] + sv[1] * Ci[4] + sv[2] * Ci[5];
            double lo = r1t[0] * Ci[6] + r1t[1] * Ci[7] + r1t[2] * Ci[8] + r1f[0] * Ci[9] + r1f[1] * Ci[10] + r1f[2] * Ci[11] -
                        (r2w[0] * Ci[12] + r2w[1] * Ci[13] + r2w[2] * Ci[14]) - (r3w[0] * Ci[15] + r3w[1] * Ci[16] + r3w[2] * Ci[17]);
            if (CT)
                lo -= cxr2[3] * Ci[18] + cxr2[4] * Ci[19] + cxr2[5] * Ci[20] + cxr3[3] * Ci[21] + cxr3[4] * Ci[22] + cxr3[5] * Ci[23];
            // branch-free select: relation bits -> 0/1 weights (columns of idle lanes are all-zero vectors)
            const double mu = (double)(unsigned)((desc_m >> i) & 1ull);   // column node i is a strict descendant of this row's node
            const double ml = (double)(unsigned)((anc_m >> i) & 1ull);    // column node i is a strict ancestor
            const double hv = mu * up + ml * lo;
            Hrow[t] = (i == lane) ? Hdiag : hv;
        }
    }
    RMX_STAMP(11)
    return Hdiag;    // H(lane,lane): the scale of this row for the solver's pivot guard
}

// One-shot evaluation (parity hook / energy kernels)
template <int NP, bool WANT_H, bool TIMED = false, bool CT = false>
__device__ __forceinline__ void eval_node(const DevModel& M, double* __restrict__ sAcc, double* __restrict__ sCol,
                                          const int lane, const double xq, const double xqd, const double xv,
                                          const double eta, NodeOut& out, double (&Hrow)[NP],
                                          unsigned long long* stamps = nullptr) {
    (void)sCol;
    FrontState fs;
    eval_front<NP, WANT_H, TIMED, CT>(M, sAcc, lane, xq, xqd, xv, eta, out, fs, stamps);
    if (WANT_H) eval_hess<NP, TIMED, CT>(M, lane, fs, Hrow, stamps, sAcc);
}

// ----------------------------------------------------------------------------- JointSpherical: Euler charts
//
// JointSpherical (matlab-diff/+redmax/JointSpherical.m) parameterises a ball joint by Euler angles in one of 12 charts,
// R = R_a1(q1) R_a2(q2) R_a3(q3) (codegen :247-262), i.e. by three revolute joints about the chart's axes: the tree holds
// such a joint as three consecutive nodes (two massless links), and everything above (kinematics, residual, Hessian) already
// covers it.  What is left is reparam_ (:63-102): after every step, a joint whose chart nears gimbal lock (|det T| <= 0.5,
// det T = +-sin q2 for the proper-Euler charts 1..6, +-cos q2 for the Tait-Bryan charts 7..12) moves to the chart with the
// largest min(|det T(R)|, |det T(R1)|), R1 the rotation of the previous step, and q, qdot (and q1, qdot1) are re-expressed.
// A chart switch swaps the K / sb constants of the group's three nodes in LDS for the new axes (DevModel::sphV).
// Charts are numbered as in the reference: 1 XYX 2 XZX 3 YZY 4 YXY 5 ZXZ 6 ZYZ 7 XYZ 8 XZY 9 YZX 10 YXZ 11 ZXY 12 ZYX.
__device__ __forceinline__ void chart_axes(const int chart, int& a1, int& a2, int& a3) {
    const int p = (chart - 1) % 6;
    a1 = p >> 1;
    a2 = (a1 + 1 + (p & 1)) % 3;
    a3 = chart <= 6 ? a1 : 3 - a1 - a2;
}
__device__ __forceinline__ double m3at(const double (&R)[9], const int i, const int j) {   // R[i][j], runtime i, j, no scratch
    double v = 0.0;
#pragma unroll
    for (int c = 0; c < 9; ++c) v = (c == 3 * i + j) ? R[c] : v;
    return v;
}
__device__ __forceinline__ void rot_elem(const int a, const double q, double (&R)[9]) {
    double sn, cs;
    sincos(q, &sn, &cs);
    const int b = (a + 1) % 3, d = (a + 2) % 3;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            R[3 * i + j] = (i == j) ? (i == a ? 1.0 : cs) : ((i == b && j == d) ? -sn : ((i == d && j == b) ? sn : 0.0));
}
__device__ __forceinline__ void mm3(const double (&A)[9], const double (&B)[9], double (&C)[9]) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
__device__ __forceinline__ void euler_R(const int chart, const double (&q)[3], double (&R)[9]) {
    int a1, a2, a3;
    chart_axes(chart, a1, a2, a3);
    double R1[9], R2[9], R3[9], T[9];
    rot_elem(a1, q[0], R1);
    rot_elem(a2, q[1], R2);
    rot_elem(a3, q[2], R3);
    mm3(R1, R2, T);
    mm3(T, R3, R);
}
// T(:,1) = R3'R2' e_a1, T(:,2) = R3' e_a2, T(:,3) = e_a3 (vee(R' dR/dq_i), :298-303); returns det T
__device__ __forceinline__ double euler_T(const int chart, const double (&q)[3], double (&T)[9]) {
    int a1, a2, a3;
    chart_axes(chart, a1, a2, a3);
    double R2[9], R3[9], t[3];
    rot_elem(a2, q[1], R2);
    rot_elem(a3, q[2], R3);
#pragma unroll
    for (int i = 0; i < 3; ++i) t[i] = m3at(R2, a1, i);                                   // R2' e_a1
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        T[3 * i + 0] = R3[i] * t[0] + R3[3 + i] * t[1] + R3[6 + i] * t[2];               // R3' t
        T[3 * i + 1] = m3at(R3, a2, i);                                                  // R3' e_a2
        T[3 * i + 2] = (i == a3) ? 1.0 : 0.0;
    }
    // det T in closed form like the reference's generated code (+-sin q2 proper Euler, +-cos q2 Tait-Bryan; only |det T| is
    // used): XYX/XZX, YZY/YXY, ZXZ/ZYZ share q2 = acos(R_ii), so their |det T| tie exactly and the first chart wins (:84).
    return (a3 == a1) ? sin(q[1]) : cos(q[1]);
}
// getEulerInv (:181-208; XYXinv..ZYXinv :1809-1949) as one rule: (i,j,k) = (a1, a2, the third axis), e = +1 when (i,j,k) is a
// cyclic permutation of (x,y,z): proper Euler q2 = acos(R_ii), q1 = atan2(R_ji, -e R_ki), q3 = atan2(R_ij, e R_ik);
// Tait-Bryan q2 = asin(e R_ik), q1 = atan2(-e R_jk, R_kk), q3 = atan2(-e R_ij, R_ii); NaN at gimbal lock.
__device__ __forceinline__ void euler_inv(const int chart, const double (&R)[9], double (&q)[3]) {
    int i, j, a3;
    chart_axes(chart, i, j, a3);
    const int k = 3 - i - j;
    const double e = ((j - i + 3) % 3 == 1) ? 1.0 : -1.0;
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    if (a3 == i) {
        const double r = m3at(R, i, i);
        const bool ok = -1.0 < r && r < 1.0;
        q[0] = ok ? atan2(m3at(R, j, i), -e * m3at(R, k, i)) : nan;
        q[1] = ok ? acos(r) : nan;
        q[2] = ok ? atan2(m3at(R, i, j), e * m3at(R, i, k)) : nan;
    } else {
        const double r = m3at(R, i, k);
        const bool ok = -1.0 < r && r < 1.0;
        q[0] = ok ? atan2(-e * m3at(R, j, k), m3at(R, k, k)) : nan;
        q[1] = ok ? asin(e * r) : nan;
        q[2] = ok ? atan2(-e * m3at(R, i, j), m3at(R, i, i)) : nan;
    }
}
// x = A\b, 3x3, Gaussian elimination with partial pivoting (MATLAB mldivide), rows swapped by value
__device__ __forceinline__ void solve3(const double (&Ain)[9], const double (&b)[3], double (&x)[3]) {
    double A[3][4];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) A[i][j] = Ain[3 * i + j];
        A[i][3] = b[i];
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
#pragma unroll
        for (int r = c + 1; r < 3; ++r) {      // bring the larger pivot candidate up (first maximum wins, as in LAPACK)
            const bool sw = fabs(A[r][c]) > fabs(A[c][c]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double u = A[c][j], v = A[r][j];
                A[c][j] = sw ? v : u;
                A[r][j] = sw ? u : v;
            }
        }
#pragma unroll
        for (int r = c + 1; r < 3; ++r) {
            const double l = A[r][c] / A[c][c];
#pragma unroll
            for (int j = c; j < 4; ++j) A[r][j] -= l * A[c][j];
        }
    }
    x[2] = A[2][3] / A[2][2];
    x[1] = (A[1][3] - A[1][2] * x[2]) / A[1][1];
    x[0] = (A[0][3] - A[0][1] * x[1] - A[0][2] * x[2]) / A[0][0];
}
// put the K / sb constants of group g's three nodes for `chart` into this wavefront's LDS copy (rows 0..41 of the constants)
template <int NP>
__device__ __forceinline__ void sph_apply_chart(const DevModel& M, double* __restrict__ sCol, const int lane, const int g, const int chart) {
    const int first = M.sph_first[g];
    const int k = lane - first;
    if (k >= 0 && k < 3) {
        int a1, a2, a3;
        chart_axes(chart, a1, a2, a3);
        const int a = k == 0 ? a1 : (k == 1 ? a2 : a3);
        const double* src = M.sphV + ((size_t)(g * 3 + k) * 3 + a) * SPH_ROWS;
        for (int r = 0; r < SPH_ROWS; ++r) sCol[r * cstride(NP) + lane] = src[r];
    }
}
// kernel prologue: bring the LDS constants in line with the stored charts (smem_setup staged CHART_XYZ)
template <int NP>
__device__ __forceinline__ void sph_setup(const DevModel& M, double* __restrict__ sCol, const int lane, const int* __restrict__ chart) {
    bool any = false;
    for (int g = 0; g < M.nsph; ++g) {
        const int c = chart[g];
        if (c != 7) {
            sph_apply_chart<NP>(M, sCol, lane, g, c);
            any = true;
        }
    }
    if (any) RMX_SYNC();
}
// Joint.reparam -> JointSpherical.reparam_ (:63-102) for every spherical group of this trajectory, after setQ at the end of a step
// (driverRedMaxBDF1.m:78, driverRedMaxBDF2.m:112).  q, qd: this lane's DOF of the new state; qp, qdp: of the previous step
// (the reference's q1, qdot1 with chart1 == chart at this point; WITH_PREV = the BDF2 driver).  The BDF1 driver never sets
// q1 / chart1, so the reference's reparam_ would fail there if a switch came up; BDF1 here picks the chart from R alone.
// Returns true if any chart changed.
template <int NP, bool WITH_PREV>
__device__ __forceinline__ bool sph_reparam(const DevModel& M, double* __restrict__ sCol, const int lane, int* __restrict__ chart,
                                            double& q, double& qd, double& qp, double& qdp) {
    bool switched = false;
    for (int g = 0; g < M.nsph; ++g) {
        const int first = M.sph_first[g];
        const int c0 = chart[g];
        const double qv[3] = {readlane_d(q, first), readlane_d(q, first + 1), readlane_d(q, first + 2)};
        double Told[9];
        const double detTold = euler_T(c0, qv, Told);
        if (fabs(detTold) > 0.5) continue;                                        // :66-68
        const double qdv[3] = {readlane_d(qd, first), readlane_d(qd, first + 1), readlane_d(qd, first + 2)};
        const double q1v[3] = {readlane_d(qp, first), readlane_d(qp, first + 1), readlane_d(qp, first + 2)};
        const double qd1v[3] = {readlane_d(qdp, first), readlane_d(qdp, first + 1), readlane_d(qdp, first + 2)};
        double R[9], R1[9], Tt[9];
        euler_R(c0, qv, R);
        if (WITH_PREV) euler_R(c0, q1v, R1);                                      // :73
        int best = 1;
        double bestv = -1.0;
        for (int k = 1; k <= 12; ++k) {                                           // :75-85
            double qk[3];
            euler_inv(k, R, qk);
            double v = fabs(euler_T(k, qk, Tt));
            v = (v == v) ? v : 0.0;
            if (WITH_PREV) {
                euler_inv(k, R1, qk);
                double v1 = fabs(euler_T(k, qk, Tt));
                v1 = (v1 == v1) ? v1 : 0.0;
                v = v1 < v ? v1 : v;
            }
            if (v > bestv) {                                                      // max() keeps the first maximum
                bestv = v;
                best = k;
            }
        }
        double w[3], Tn[9], qn[3], qdn[3], q1n[3] = {0, 0, 0}, qd1n[3] = {0, 0, 0};
#pragma unroll
        for (int i = 0; i < 3; ++i) w[i] = Told[3 * i] * qdv[0] + Told[3 * i + 1] * qdv[1] + Told[3 * i + 2] * qdv[2];
        euler_inv(best, R, qn);                                                   // :87
        euler_T(best, qn, Tn);                                                    // :89
        solve3(Tn, w, qdn);                                                       // :91
        if (WITH_PREV) {                                                          // :97-101
            euler_T(c0, q1v, Told);
#pragma unroll
            for (int i = 0; i < 3; ++i) w[i] = Told[3 * i] * qd1v[0] + Told[3 * i + 1] * qd1v[1] + Told[3 * i + 2] * qd1v[2];
            euler_inv(best, R1, q1n);
            euler_T(best, q1n, Tn);
            solve3(Tn, w, qd1n);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (lane == first + k) {
                q = qn[k];
                qd = qdn[k];
                if (WITH_PREV) {
                    qp = q1n[k];
                    qdp = qd1n[k];
                }
            }
        if (best != c0) {
            if (lane == 0) chart[g] = best;
            sph_apply_chart<NP>(M, sCol, lane, g, best);
            switched = true;
        }
    }
    if (switched) RMX_SYNC();
    return switched;
}

// ----------------------------------------------------------------------------- dense solve
//
// dx = -H\g (driverRedMaxBDF1.m:117: MATLAB mldivide = LU with partial pivoting).  Lane = row; the row
// lives in registers; rows are never moved (implicit permutation); the pivot row is broadcast with
// v_readlane into scalar registers; the right-hand side is eliminated alongside.
// Fast path: eliminate on the diagonal (no search, static lanes) under a growth guard.  Every multiplier of the symmetrically
// equilibrated matrix D^-1/2 H D^-1/2 (D = diag H) must satisfy |l'| <= LU_GROWTH_MAX, i.e. the diagonal passes threshold
// pivoting with tau = 1/LU_GROWTH_MAX (the criterion sparse direct solvers use) on the scaled matrix, and every pivot must be
// positive: l'_ik^2 = l_ik^2 u_kk / d_i <= 64.  For a symmetric positive definite matrix l' <= 1 always (a_ik^2 <= a_ii a_kk
// on every Schur complement), and H = M - eta D - eta^2 K + dMdq.v is a modest perturbation of the SPD mass matrix, so this
// nearly always holds - also for trees that mix prismatic (mass-sized diagonal) and revolute (inertia-sized) DOFs, where the
// raw multipliers are large for scaling reasons only.  If any step violates the guard (or is NaN) `ok` comes back false and
// the caller redoes the solve with full partial pivoting (lu_solve_neg) on a re-assembled H: pivoting semantics are kept,
// its cost is paid only when needed.
#ifndef RMX_BACKSUB32
#define RMX_BACKSUB32 1            // 1: back substitution without lane conditions (v_writelane capture); 0: the select form (build variants)
#endif
constexpr double LU_GROWTH_MAX = 8.0;
constexpr int LU_BATCH = 8;
// The positive-pivot half of the guard: every pivot must be positive, finite and non-zero.  It is tested on the reciprocals, which
// the solves compute anyway, with ONE integer instruction per pivot: as unsigned integers the high words of doubles order as
// positive finite < +inf, NaN (0x7ff00000 ...) < negative (0x80000000 ...), so the running unsigned maximum of the high words
// stays below 0x7ff00000 exactly when every reciprocal is positive and finite (a zero or denormal pivot gives inf / NaN).
// (fmin on the pivots cost two instructions each: the compiler canonicalises the scalar operand of v_min_f64 first.)
#ifndef RMX_PIVGUARD_INT
#define RMX_PIVGUARD_INT 1
#endif
// The growth half of the guard, l'^2 = l^2 u_kk / d_i <= LU_GROWTH_MAX^2 with l^2 u_kk = a_ik l: the running maximum of the products
// a_ik l is kept on their HIGH WORDS as signed integers (for non-negative doubles integer order is numeric order; a negative product
// or -0 - a row that is not being eliminated has l = 0 - compares low and is ignored: negative products need a negative pivot, which
// PivGuard reports; a NaN with a clear sign bit compares high and trips the guard, where v_max_f64 would drop it).  One integer
// instruction per update: fmax cost two once the accumulators are pinned (an empty asm makes the compiler canonicalise its output
// before the next v_max_f64), and a threshold that is a heuristic anyway loses nothing to the 2^-20 granularity of the high word.
struct GrowGuard {
    int hi = 0;
    __device__ __forceinline__ void see(const double prod) {
#ifdef RMX_VAR_NO_GROW_GUARD      // measurement aid (tools/build_variant.py): what the growth half of the guard costs
        (void)prod;
#else
        const int h = __double2hiint(prod);
        hi = h > hi ? h : hi;
#endif
    }
    __device__ __forceinline__ void pin() { asm volatile("" : "+v"(hi)); }
    __device__ __forceinline__ bool bad(const double lim) const { return hi > __double2hiint(lim); }      // lim = 64 d_i > 0
};
struct PivGuard {
#if RMX_PIVGUARD_INT
    unsigned hi = 0u;
    __device__ __forceinline__ void see(const double piv, const double rinv) {
        (void)piv;
        const unsigned h = (unsigned)__double2hiint(rinv);
        hi = h > hi ? h : hi;
    }
    __device__ __forceinline__ void pin() { asm volatile("" : "+v"(hi)); }
    __device__ __forceinline__ bool ok() const { return hi < 0x7ff00000u; }
#else
    double pmin = 1.0;
    __device__ __forceinline__ void see(const double piv, const double rinv) {
        (void)rinv;
        pmin = fmin(pmin, piv);
    }
    __device__ __forceinline__ void pin() { asm volatile("" : "+v"(pmin)); }
    __device__ __forceinline__ bool ok() const { return pmin > 0.0; }
#endif
};
__device__ __forceinline__ void lu_pin(double (&pv)[LU_BATCH]) {
    asm volatile("" : "+s"(pv[0]), "+s"(pv[1]), "+s"(pv[2]), "+s"(pv[3]), "+s"(pv[4]), "+s"(pv[5]), "+s"(pv[6]), "+s"(pv[7]));
}
// The last 16 pivots of lu_solve_neg_diag (K is a template constant: the DPP lane is an immediate).  Every row still being
// eliminated sits in the pivot row's own 16-lane DPP row, so the pivot-row broadcast rides on the FMA (fmsub_rowbcast).  Rows
// in the other DPP rows are finished (l == 0: they add 0 x a finite entry of one of their own finished rows) or idle mirrors.
template <int NP, int K, int NR>
__device__ __forceinline__ void lu_diag_tail(double (&Hrow)[NP], double& b, GrowGuard& gmax, PivGuard& pg, double& piv, double& rinv,
                                             double (&rinvs)[NR], double& rinv_own, const int lv) {
    if constexpr (K < NP) {
        constexpr int N = K - (NP - 16);
        if constexpr (NR == NP) rinvs[K] = rinv;
        else rinv_own = (lv == K) ? rinv : rinv_own;
        const double l = (lv > K) ? Hrow[K] * rinv : 0.0;
        gmax.see(Hrow[K] * l);
        pg.see(piv, rinv);
        constexpr bool LAST = K + 2 >= NP;      // too few instructions left in a step to separate dependent broadcasts
        if constexpr (K + 1 < NP) {
            fmsub_rowbcast<N, LAST>(Hrow[K + 1], Hrow[K + 1], l);
            piv = readlane_d(Hrow[K + 1], K + 1);
            rinv = recip(piv);
        }
#pragma unroll
        for (int c = K + 2; c < NP; ++c) fmsub_rowbcast<N>(Hrow[c], Hrow[c], l);
        fmsub_rowbcast<N, LAST>(b, b, l);
        lu_diag_tail<NP, K + 1, NR>(Hrow, b, gmax, pg, piv, rinv, rinvs, rinv_own, lv);
    }
}

template <int NP>
__device__ __forceinline__ double lu_solve_neg_diag(const int lane, double (&Hrow)[NP], const double g, const double diag_own,
                                                    bool& ok) {
    double b = -g;
    // guard state: the largest scaled multiplier seen by this lane and the smallest pivot (wave-uniform), compared once at the
    // end (v_max / v_min per step instead of two compares and two mask updates; a NaN passes through v_max but shows up in dx,
    // which the caller checks)
    GrowGuard gmax;
    PivGuard pg;
    // The lane comparisons below (lane > k, lane == k, lane < k for 32..64 values of k) are invariant across Newton iterations;
    // hoisted out of the loops they would be ~100 64-bit masks held in SGPRs, spilled to VGPR lanes and fetched back with
    // v_readlane inside the elimination, and the starved allocator would serialise the pivot-row broadcasts.  An opaque copy
    // of the lane id keeps the compares (one VALU instruction each) where they are used.
    int lv = lane;
    asm volatile("" : "+v"(lv));
    // 1/U(k,k) for the back substitution: n <= 32 keeps all of them (wave-uniform values, no per-lane select per step);
    // 64 rows would cost 128 more registers, so there every lane keeps its own
    constexpr bool KEEP_ALL = NP <= 32;
    double rinv_own = 0.0;
    double rinvs[KEEP_ALL ? NP : 1];
    const double lim = (LU_GROWTH_MAX * LU_GROWTH_MAX) * diag_own;
    // the reciprocal of pivot k+1 is started as soon as column k+1 has seen step k, ahead of the other trailing columns, so
    // that its latency chain (readlane -> rcp -> 2 Newton steps) overlaps their updates
    double piv = readlane_d(Hrow[0], 0);
    double rinv = recip(piv);
    constexpr int KTAIL = (NP >= 16 && LU_DPP_TAIL) ? NP - 16 : NP;   // pivots from KTAIL on: lu_diag_tail
#pragma unroll
    for (int k = 0; k < KTAIL; ++k) {
        if constexpr (KEEP_ALL) rinvs[k] = rinv;
        else rinv_own = (lv == k) ? rinv : rinv_own;
        const double l = (lv > k) ? Hrow[k] * rinv : 0.0;
        gmax.see(Hrow[k] * l);                   // l^2 u_kk = a_ik l
        pg.see(piv, rinv);
        if (k + 1 < NP) {
            Hrow[k + 1] -= l * readlane_d(Hrow[k + 1], k);
            piv = readlane_d(Hrow[k + 1], k + 1);
            rinv = recip(piv);
        }
        // The pivot row is broadcast LU_BATCH entries at a time, all of a batch before its first FMA: an FMA that reads the scalar
        // registers a v_readlane has just written needs hazard wait states, and the element-by-element order costs 22 cycles
        // per element against 15 for batches of 8 (tools/ubench.hip).  The "+s" pin keeps the scheduler from re-interleaving.
        if constexpr (NP > 32) {    // 64 rows: batches of 8 push the register allocator over the edge (10x slower); batches of 4
#pragma unroll
            for (int c0 = k + 2; c0 < NP; c0 += 4) {
                double pv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) pv[i] = (c0 + i < NP) ? readlane_d(Hrow[c0 + i < NP ? c0 + i : NP - 1], k) : 0.0;
                asm volatile("" : "+s"(pv[0]), "+s"(pv[1]), "+s"(pv[2]), "+s"(pv[3]));
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (c0 + i < NP) Hrow[c0 + i] -= l * pv[i];
            }
        } else
#pragma unroll
        for (int c0 = k + 2; c0 < NP; c0 += LU_BATCH) {
            double pv[LU_BATCH];
#pragma unroll
            for (int i = 0; i < LU_BATCH; ++i) pv[i] = (c0 + i < NP) ? readlane_d(Hrow[c0 + i < NP ? c0 + i : NP - 1], k) : 0.0;
            lu_pin(pv);
#pragma unroll
            for (int i = 0; i < LU_BATCH; ++i)
                if (c0 + i < NP) Hrow[c0 + i] -= l * pv[i];
        }
        b -= l * readlane_d(b, k);
    }
    if constexpr (KTAIL < NP) lu_diag_tail<NP, KTAIL>(Hrow, b, gmax, pg, piv, rinv, rinvs, rinv_own, lv);
    double dx = 0.0;
    if constexpr (KEEP_ALL && RMX_BACKSUB32 == 1) {
        // without lane conditions: see lu_solve_neg_diag32 (x_k written into lane k of dx, the update on all lanes)
        int dlo = 0, dhi = 0;
#pragma unroll
        for (int k = NP - 1; k >= 0; --k) {
            const double t = b * rinvs[k];
            const int slo = __builtin_amdgcn_readlane(__double2loint(t), k), shi = __builtin_amdgcn_readlane(__double2hiint(t), k);
            dlo = writelane_i(slo, k, dlo);
            dhi = writelane_i(shi, k, dhi);
            if (k > 0) b = fma(-Hrow[k], __hiloint2double(shi, slo), b);
        }
        dx = __hiloint2double(dhi, dlo);
    } else {
#pragma unroll
        for (int k = NP - 1; k >= 0; --k) {
            double xk;
            if constexpr (KEEP_ALL) xk = readlane_d(b, k) * rinvs[k];
            else xk = readlane_d(b * rinv_own, k);
            if (lv == k) dx = xk;
            if (lv < k) b -= Hrow[k] * xk;
        }
    }
    // lanes beyond the padded size may carry mirrored rows (eval_hess ZERO_IDLE = false): their guard is ignored
    ok = !__any(lane < NP && gmax.bad(lim)) && pg.ok();
    return dx;
}

// ---- n <= 32, H staged row-major in LDS by the MFMA Hessian: pivots 0..15 without a single v_readlane broadcast.
// Every 16-lane DPP row r = lane >> 4 works on ALL 32 matrix rows, two per lane (set A: row j = lane & 15, set B: row 16 + j),
// and holds columns 0..15 (replicated in the four DPP rows: they are the pivot columns of this phase, so every DPP row can
// form the multipliers itself) plus its own four of the columns 16..31 (16 + 4 r ...).  The pivot row k < 16 is set A of lane k
// of the SAME DPP row, so every update is one v_fmac_f64_dpp (fmsub_rowbcast): 2 (15 - k) + 8 + 2 of them per pivot, 400 in
// all, against 376 x (2 v_readlane + FMA) plus the right-hand side in the row-per-lane layout.  The four column quarters then
// return through LDS to row-per-lane (lane = row), which is exactly the state lu_solve_neg_diag has after 16 pivots: the last
// 16 pivots (lu_diag_tail) and the back substitution are shared.  Same operations on the same values in the same order
// per matrix entry: the results are bit-identical to lu_solve_neg_diag.
#ifndef RMX_P1_BFIRST
#define RMX_P1_BFIRST 1
#endif
template <int K>
__device__ __forceinline__ void lu32_phase1(double (&A)[16], double (&AX)[4], double (&B)[16], double (&BX)[4], double& bA,
                                            double& bB, GrowGuard& gmaxA, GrowGuard& gmaxB, PivGuard& pg, double& piv, double& rinv,
                                            double (&rinvs)[32], const int jv) {
    if constexpr (K < 16) {
        rinvs[K] = rinv;
        const double lA = (jv > K) ? A[K] * rinv : 0.0;
        const double lB = B[K] * rinv;
        gmaxA.see(A[K] * lA);
        gmaxB.see(B[K] * lB);
        pg.see(piv, rinv);
        // pinned where they are computed: left alone, a third of these guard updates sink out of the elimination with their
        // operands parked in AGPRs (400 v_accvgpr moves in the kernel): 4.67 -> 4.47 ms per 100 steps of the 32-chain
        gmaxA.pin();
        gmaxB.pin();
        pg.pin();
#if RMX_P1_BFIRST
        // Order inside a step: every update of set B (rows 16..31) reads the pivot row out of a set-A register (lane K, which the
        // step's own set-A update leaves unchanged: its multiplier is 0), so B goes FIRST and A second.  With A first, each B
        // update followed the asm statement that had just written its source register, and the compiler - which cannot see into
        // the asm - put a wait state between every such pair (gfx950's conservative forwarding hazard for inline asm): ~4 s_nop
        // per pivot, each a full issue slot of the lone wavefront.  The arithmetic per entry is the same either way.
        if constexpr (K + 1 < 16) {
            // the pivot columns are replicated in the four DPP rows: the next pivot is lane K + 1 of the lane's own row, one
            // v_mov_b64_dpp instead of two v_readlane plus the wait states of the scalar round trip (the reciprocal's operand)
            fmsub_rowbcast<K>(B[K + 1], A[K + 1], lB);
            fmsub_rowbcast<K>(A[K + 1], A[K + 1], lA);
            piv = row_bcast<K + 1>(A[K + 1]);
            rinv = recip(piv);
        } else {                       // pivot 16 is row 16 (set B of lane 0), column 16 (first extra column of DPP row 0)
            fmsub_rowbcast<K>(BX[0], AX[0], lB);
            piv = readlane_d(BX[0], 0);
            rinv = recip(piv);
        }
#pragma unroll
        for (int c = K + 2; c < 16; ++c) fmsub_rowbcast<K>(B[c], A[c], lB);
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (K + 1 < 16 || c > 0) fmsub_rowbcast<K>(BX[c], AX[c], lB);
        fmsub_rowbcast<K>(bB, bA, lB);
#pragma unroll
        for (int c = K + 2; c < 16; ++c) fmsub_rowbcast<K>(A[c], A[c], lA);
#pragma unroll
        for (int c = 0; c < 4; ++c) fmsub_rowbcast<K>(AX[c], AX[c], lA);
        fmsub_rowbcast<K>(bA, bA, lA);
#else      // the order of rounds 1-3 (build variants)
        if constexpr (K + 1 < 16) {
            fmsub_rowbcast<K>(A[K + 1], A[K + 1], lA);
            piv = row_bcast<K + 1>(A[K + 1]);
            rinv = recip(piv);
        } else {
            fmsub_rowbcast<K>(BX[0], AX[0], lB);
            piv = readlane_d(BX[0], 0);
            rinv = recip(piv);
        }
#pragma unroll
        for (int c = K + 2; c < 16; ++c) fmsub_rowbcast<K>(A[c], A[c], lA);
#pragma unroll
        for (int c = K + 1; c < 16; ++c) fmsub_rowbcast<K>(B[c], A[c], lB);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            fmsub_rowbcast<K>(AX[c], AX[c], lA);
            if (K + 1 < 16 || c > 0) fmsub_rowbcast<K>(BX[c], AX[c], lB);
        }
        fmsub_rowbcast<K>(bB, bA, lB);
        fmsub_rowbcast<K>(bA, bA, lA);
#endif
        lu32_phase1<K + 1>(A, AX, B, BX, bA, bB, gmaxA, gmaxB, pg, piv, rinv, rinvs, jv);
    }
}

__device__ __forceinline__ double lu_solve_neg_diag32(const int n, const int lane, double* sAcc, const double g, bool& ok) {
    constexpr int NP = 32;
    typedef double v2d __attribute__((ext_vector_type(2)));
    double* sH = sAcc;                                   // H, row-major [32][HM_H_STRIDE]; columns 32, 33 of a row are spare
    const int r4 = lane >> 4, j = lane & 15;
    (void)g;                                             // eval_hess has staged -g in column 32 of the rows
    double A[16], AX[4], B[16], BX[4];
    const double* rowA = sH + j * HM_H_STRIDE;
    const double* rowB = sH + (16 + j) * HM_H_STRIDE;
    {
        const v2d* ra = reinterpret_cast<const v2d*>(rowA);
        const v2d* rb = reinterpret_cast<const v2d*>(rowB);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const v2d ta = ra[c], tb = rb[c];
            A[2 * c] = ta[0];
            A[2 * c + 1] = ta[1];
            B[2 * c] = tb[0];
            B[2 * c + 1] = tb[1];
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const v2d ta = ra[8 + 2 * r4 + c], tb = rb[8 + 2 * r4 + c];
            AX[2 * c] = ta[0];
            AX[2 * c + 1] = ta[1];
            BX[2 * c] = tb[0];
            BX[2 * c + 1] = tb[1];
        }
    }
    double bA = rowA[32], bB = rowB[32];
    const double limA = (LU_GROWTH_MAX * LU_GROWTH_MAX) * rowA[j], limB = (LU_GROWTH_MAX * LU_GROWTH_MAX) * rowB[16 + j];
    int jv = j, lv = lane;                               // opaque copies: see lu_solve_neg_diag
    asm volatile("" : "+v"(jv), "+v"(lv));
    GrowGuard gmaxA, gmaxB;
    PivGuard pg;
    double rinvs[NP];
    double piv = row_bcast<0>(A[0]);
    double rinv = recip(piv);
    lu32_phase1<0>(A, AX, B, BX, bA, bB, gmaxA, gmaxB, pg, piv, rinv, rinvs, jv);
    // the column quarters 16 + 4 r .. of both row sets go back to their rows; lane = row reads columns 16..31
    RMX_SYNC();
    {
        v2d* wa = reinterpret_cast<v2d*>(sH + j * HM_H_STRIDE + 16 + 4 * r4);
        v2d* wb = reinterpret_cast<v2d*>(sH + (16 + j) * HM_H_STRIDE + 16 + 4 * r4);
        wa[0] = v2d{AX[0], AX[1]};
        wa[1] = v2d{AX[2], AX[3]};
        wb[0] = v2d{BX[0], BX[1]};
        wb[1] = v2d{BX[2], BX[3]};
    }
    RMX_SYNC();
    double Hrow[NP];
#pragma unroll
    for (int c = 0; c < 16; ++c) Hrow[c] = A[c];         // rows 0..15 (lanes 0..15); never read on the other lanes
    {
        const v2d* hr = reinterpret_cast<const v2d*>(sH + (lane & 31) * HM_H_STRIDE + 16);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const v2d t = hr[c];
            Hrow[16 + 2 * c] = t[0];
            Hrow[16 + 2 * c + 1] = t[1];
        }
    }
    const bool rowsA = (lane & 16) == 0;                 // lanes 0..15 (and their idle mirrors 32..47) carry set A
    double b = rowsA ? bA : bB;
    GrowGuard gmax;
    gmax.hi = rowsA ? gmaxA.hi : gmaxB.hi;
    const double lim = rowsA ? limA : limB;
    double rinv_own = 0.0;
    lu_diag_tail<NP, 16>(Hrow, b, gmax, pg, piv, rinv, rinvs, rinv_own, lv);
    double dx = 0.0;
#if RMX_BACKSUB32 == 1
    // x_k = b_k / U_kk is formed in every lane (lane k holds it), read out of lane k into a scalar pair and written into lane k of dx
    // (v_writelane); the update b_i -= U_ik x_k then runs on ALL lanes: a lane i >= k is finished (its b_i was consumed at step i,
    // k descends) and what it computes from its stale b_i is never read.  5-6 issue slots per step and no lane condition, against
    // compare + wait state + two selects for the update and another compare + two selects for the capture of dx (the compiler sank
    // the latter into a 134-instruction select chain behind the solve).  Same products, same sums: bit-identical.
    {
        int dlo = 0, dhi = 0;
#pragma unroll
        for (int k = NP - 1; k >= 0; --k) {
            const double t = b * rinvs[k];
            const int slo = __builtin_amdgcn_readlane(__double2loint(t), k), shi = __builtin_amdgcn_readlane(__double2hiint(t), k);
            dlo = writelane_i(slo, k, dlo);
            dhi = writelane_i(shi, k, dhi);
            if (k > 0) b = fma(-Hrow[k], __hiloint2double(shi, slo), b);
        }
        dx = __hiloint2double(dhi, dlo);
    }
#else
#pragma unroll
    for (int k = NP - 1; k >= 0; --k) {
        const double xk = readlane_d(b, k) * rinvs[k];
        if (lv == k) dx = xk;
        if (lv < k) b -= Hrow[k] * xk;
    }
#endif
    ok = !__any(lane < NP && gmax.bad(lim)) && pg.ok();
    RMX_SYNC();                 // sAcc goes back to the front, whose subtree scan relies on a zero row n
    if (lane < ACC_STRIDE) sAcc[n * ACC_STRIDE + lane] = 0.0;
    RMX_SYNC();
    return dx;
}

// ---- 33..64 rows, one wavefront: the whole guarded solve without a single v_readlane broadcast of a pivot row.
// H lives row-major in LDS (sH, the front's scratch) and is eliminated in four phases of 16 pivots.  In phase P every 16-lane
// DPP row works on ALL rows still being eliminated, one row per lane and row set (set s: row 16 s + j, j = lane & 15; sets < P
// are finished), so the pivot row 16 P + K is set P of lane K of the lane's OWN DPP row and every update is one v_fmac_f64_dpp
// (fmsub_rowbcast).  Pass A eliminates inside the 16 pivot columns of the phase (S, replicated in the four DPP rows, so each forms
// the multipliers itself) and leaves the multipliers in place; pass B applies them, pivot by pivot in the same order, to the
// later column blocks, each DPP row taking its quarter of a block's columns through registers and back to LDS.  The back
// substitution runs in the same layout, block by block (the value x_k rides on the DPP broadcast).  Every matrix entry sees the
// same operations on the same values in the same order as in lu_solve_neg_diag<64>: the results are bit-identical.
template <int P, int K, bool PINMIN>
__device__ __forceinline__ void lu64_pass_a(double (&S)[4][16], double (&b)[4], GrowGuard (&gm)[4], double (&rown)[4], PivGuard& pg,
                                            double& piv, double& rinv, const int jv) {
    if constexpr (K < 16) {
        rown[P] = (jv == K) ? rinv : rown[P];
        double l[4] = {0.0, 0.0, 0.0, 0.0};
        l[P] = (jv > K) ? S[P][K] * rinv : 0.0;
#pragma unroll
        for (int s = P + 1; s < 4; ++s) l[s] = S[s][K] * rinv;
#pragma unroll
        for (int s = P; s < 4; ++s) {
            gm[s].see(S[s][K] * l[s]);
            // pinned here: left alone, the compiler sinks all 230 guard updates of a solve to its end and keeps their operands
            // alive (in scratch) until then
            gm[s].pin();
        }
        pg.see(piv, rinv);
        // (same reason: 64 guard updates and their pivots parked in SGPR spills until the end)
        if constexpr (PINMIN) pg.pin();
        if constexpr (K + 1 < 16) {
            fmsub_rowbcast<K>(S[P][K + 1], S[P][K + 1], l[P]);
            piv = row_bcast<K + 1>(S[P][K + 1]);       // replicated pivot columns: see lu32_phase1
            rinv = recip(piv);
        }
        // the multipliers stay where the column was (rows at or above the pivot keep their U entries)
        S[P][K] = (jv > K) ? l[P] : S[P][K];
#pragma unroll
        for (int s = P + 1; s < 4; ++s) S[s][K] = l[s];
        // the later sets first, the pivot set's own rows last: a broadcast then never reads a register that one of the two
        // preceding instructions wrote (lane K of the pivot set's rows is not changed by its own update - its multiplier is 0 -
        // so the order does not change a bit, but the hazard costs a wait state each time)
#pragma unroll
        for (int s = P + 1; s < 4; ++s)
#pragma unroll
            for (int c = K + 1; c < 16; ++c) fmsub_rowbcast<K>(S[s][c], S[P][c], l[s]);
#pragma unroll
        for (int c = K + 2; c < 16; ++c) fmsub_rowbcast<K>(S[P][c], S[P][c], l[P]);
#pragma unroll
        for (int s = P + 1; s < 4; ++s) fmsub_rowbcast<K>(b[s], b[P], l[s]);
        fmsub_rowbcast<K>(b[P], b[P], l[P]);
        lu64_pass_a<P, K + 1, PINMIN>(S, b, gm, rown, pg, piv, rinv, jv);
    }
}
template <int P, int K, int CW>
__device__ __forceinline__ void lu64_pass_b(const double (&S)[4][16], double (&X)[4][CW], const int jv) {
    if constexpr (K < 16) {
        const double lP = (jv > K) ? S[P][K] : 0.0;
#pragma unroll
        for (int s = P + 1; s < 4; ++s)
#pragma unroll
            for (int c = 0; c < CW; ++c) fmsub_rowbcast<K>(X[s][c], X[P][c], S[s][K]);
#pragma unroll
        for (int c = 0; c < CW; ++c) fmsub_rowbcast<K>(X[P][c], X[P][c], lP);
        lu64_pass_b<P, K + 1, CW>(S, X, jv);
    }
}
// One phase of 16 pivots.
template <int P>
__device__ __forceinline__ void lu64_phase(double* sH, const int lane, double (&b)[4], GrowGuard (&gm)[4], double (&rown)[4],
                                           PivGuard& pg, const int jv) {
    typedef double v2d __attribute__((ext_vector_type(2)));
    constexpr int CW = RMX_W2 ? 2 : 4;         // columns of a later block per DPP row (RMX_W2: the 8 DPP rows of two waves)
    const int r4 = lane >> 4, j = lane & 15;
    const int r8 = RMX_W2 ? 4 * (int)(threadIdx.x >> 6) + r4 : r4;
    double S[4][16];
#pragma unroll
    for (int s = P; s < 4; ++s) {
        const v2d* rd = reinterpret_cast<const v2d*>(sH + (16 * s + j) * H64_STRIDE + 16 * P);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const v2d t = rd[c];
            S[s][2 * c] = t[0];
            S[s][2 * c + 1] = t[1];
        }
    }
    double piv = row_bcast<0>(S[P][0]);
    double rinv = recip(piv);
    lu64_pass_a<P, 0, true>(S, b, gm, rown, pg, piv, rinv, jv);
#pragma unroll
    for (int B = P + 1; B < 4; ++B) {
        __builtin_amdgcn_sched_barrier(0);      // one block in registers at a time (hoisted loads of all blocks spill)
        double X[4][CW];
        const int col = 16 * B + CW * r8;
#pragma unroll
        for (int s = P; s < 4; ++s) {
            const v2d* rd = reinterpret_cast<const v2d*>(sH + (16 * s + j) * H64_STRIDE + col);
#pragma unroll
            for (int c = 0; c < CW / 2; ++c) {
                const v2d t = rd[c];
                X[s][2 * c] = t[0];
                X[s][2 * c + 1] = t[1];
            }
        }
        lu64_pass_b<P, 0, CW>(S, X, jv);
#pragma unroll
        for (int s = P; s < 4; ++s) {
            v2d* w = reinterpret_cast<v2d*>(sH + (16 * s + j) * H64_STRIDE + col);
#pragma unroll
            for (int c = 0; c < CW / 2; ++c) w[c] = v2d{X[s][2 * c], X[s][2 * c + 1]};
        }
    }
    if constexpr (RMX_W2 && (P < 3 || RMX_W2_FULL_HELPER)) RMX_WG_BAR();      // (phase 3 has no later block: wave 0 alone)
    else RMX_SYNC();                           // the next phase reads what all the DPP rows have written
    // the finished rows of this phase: their part of U, for the back substitution
    if (RMX_W2 ? threadIdx.x < 16 : lane < 16) {
        v2d* w = reinterpret_cast<v2d*>(sH + (16 * P + j) * H64_STRIDE + 16 * P);
#pragma unroll
        for (int c = 0; c < 8; ++c) w[c] = v2d{S[P][2 * c], S[P][2 * c + 1]};
    }
}
// back substitution inside the diagonal block of set P: x_k = b_k / U(k,k), rows above take U(j,k) x_k off (k = 15 .. 0)
template <int K>
__device__ __forceinline__ void lu64_back_diag(double& bP, const double rP, const double (&U)[16], const int jv) {
    if constexpr (K >= 0) {
        const double xk = bP * rP;                     // lane K: x_K (its updates from the larger k are complete)
        const double m = (jv < K) ? U[K] : 0.0;
        fmsub_rowbcast<K, true>(bP, xk, m);
        lu64_back_diag<K - 1>(bP, rP, U, jv);
    }
}

// rows of an earlier set take the finished block's x off: b(row) -= U(row, 16 P + k) x_k, k = 15 .. 0 (the order of the
// row-per-lane back substitution)
template <int K>
__device__ __forceinline__ void lu64_back_off(double& bs, const double xP, const double (&T)[16]) {
    if constexpr (K >= 0) {
        fmsub_rowbcast<K>(bs, xP, T[K]);
        lu64_back_off<K - 1>(bs, xP, T);
    }
}

// H and the right-hand side are in place: sAcc = [64][H64_STRIDE], row-major, column 64 = -g
__device__ __forceinline__ double lu_solve_neg_diag64_staged(const int n, const int lane, double* sAcc, bool& ok) {
    typedef double v2d __attribute__((ext_vector_type(2)));
    double* sH = sAcc;                                   // the front's scratch is free during the solve
    const int r4 = lane >> 4, j = lane & 15;
    int jv = j;                                          // opaque copy: see lu_solve_neg_diag
    asm volatile("" : "+v"(jv));
    double b[4], lim[4], rown[4] = {0.0, 0.0, 0.0, 0.0};
    GrowGuard gm[4];
    PivGuard pg;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const double* row = sH + (16 * s + j) * H64_STRIDE;
        b[s] = row[64];
        lim[s] = (LU_GROWTH_MAX * LU_GROWTH_MAX) * row[16 * s + j];
    }
    lu64_phase<0>(sH, lane, b, gm, rown, pg, jv);
    lu64_phase<1>(sH, lane, b, gm, rown, pg, jv);
    lu64_phase<2>(sH, lane, b, gm, rown, pg, jv);
    if constexpr (RMX_W2 && !RMX_W2_FULL_HELPER) {     // the helper wave has applied its share of the last later block: the rest is wave 0's
        if (threadIdx.x >= 64) {
            ok = true;
            return 0.0;
        }
    }
    lu64_phase<3>(sH, lane, b, gm, rown, pg, jv);
    if constexpr (RMX_W2 && RMX_W2_FULL_HELPER) RMX_WG_BAR();
    else RMX_SYNC();            // (the finished rows of phase 3)
    // back substitution, block column by block column from the right
    double x[4];
#pragma unroll
    for (int P = 3; P >= 0; --P) {
        double U[16];
        {
            const v2d* rd = reinterpret_cast<const v2d*>(sH + (16 * P + j) * H64_STRIDE + 16 * P);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const v2d t = rd[c];
                U[2 * c] = t[0];
                U[2 * c + 1] = t[1];
            }
        }
        lu64_back_diag<15>(b[P], rown[P], U, jv);
        x[P] = b[P] * rown[P];
#pragma unroll
        for (int s = 0; s < 4; ++s)
            if (s < P) {
                const v2d* rd = reinterpret_cast<const v2d*>(sH + (16 * s + j) * H64_STRIDE + 16 * P);
                double T[16];
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const v2d t = rd[c];
                    T[2 * c] = t[0];
                    T[2 * c + 1] = t[1];
                }
                lu64_back_off<15>(b[s], x[P], T);
            }
    }
    bool bad = false;
#pragma unroll
    for (int s = 0; s < 4; ++s) bad = bad || gm[s].bad(lim[s]);
    ok = !__any(bad) && pg.ok();
    const double dx = r4 == 0 ? x[0] : (r4 == 1 ? x[1] : (r4 == 2 ? x[2] : x[3]));
    if constexpr (RMX_W2) {     // (the caller hands the scratch back to the front: w2_lu_call is one function for every n)
        if constexpr (RMX_W2_FULL_HELPER) RMX_WG_BAR();           // both waves have read their last of H
        return dx;
    }
    RMX_SYNC();                 // sAcc goes back to the front, whose subtree scan relies on a zero row n
    if (lane < ACC_STRIDE) sAcc[n * ACC_STRIDE + lane] = 0.0;
    RMX_SYNC();
    return dx;
}

// ---- 33..64 rows of a BRANCHING tree: the guarded solve as a multifrontal elimination along the tree.
// H(i, j) is non-zero only where one of the nodes i, j is an ancestor of the other (the Hessian stage's relation masks), so in
// leaves-first order the elimination creates no fill outside the root paths: node i owns a frontal matrix over itself and its
// ancestors, G[k1][k2] with k = levels above i (0 = i itself) - row 0 = H(i, ancestors), column 0 = H(ancestors, i), the rest the
// updates its subtree has accumulated for its ancestors - and the right-hand side rides along as one more column.  Lane = node; ALL
// nodes of a depth level are eliminated at once (they share no unknown), then every parent adds its children's update matrices
// (shifted by one level, handed over through the scratch) to its own, children in listing order: depth rounds of at most
// (d + 1)(d + 2) fused multiply-adds per node instead of 64 pivot steps over 64 x 64 - the 64-joint tree of configs[2] (depth 6,
// 592 of 4096 entries non-zero): ~1.1 k instructions and 6 dependent levels against 4 k instructions and 64 dependent pivots.  The
// back substitution walks down: a node reads its ancestors' solutions from the scratch.  The entries of H themselves are formed here,
// from the Hessian stage's staged row and column vectors (2 x depth dot products per node): no H matrix exists on this path.
// Guards as in the dense solve (every pivot positive and finite,
// every multiplier of the equilibrated matrix below LU_GROWTH_MAX); a tripped guard sends the caller to the pivoting dense solve.
// Not the dense solve's elimination order, hence not its rounding: results agree with it to roundoff, not bit for bit.
struct TreeLane {
    int depth, parent;
    int up[TREE_DMAX + 1];       // up[k]: the ancestor k levels up (k = 1 .. depth), -1 beyond the root
    int child[TREE_CMAX];
};
__device__ __forceinline__ TreeLane tree_lane(const DevModel& M, const int lane) {
    TreeLane t;
    const int* T = M.tree + lane;
    t.depth = T[0];
    t.parent = T[(1 + TREE_DMAX + TREE_CMAX) * MAXN];
#pragma unroll
    for (int c = 0; c < TREE_CMAX; ++c) t.child[c] = T[(1 + TREE_DMAX + c) * MAXN];
    t.up[0] = lane;
#pragma unroll
    for (int k = 1; k <= TREE_DMAX; ++k) t.up[k] = T[k * MAXN];      // (-1 beyond the root; every load independent of the others)
    return t;
}
constexpr int TREE_NR = TREE_DMAX + 1, TREE_RC = TREE_DMAX + 1;      // rows / columns of a frontal matrix (levels 0 .. TREE_DMAX), its right-hand-side column

// level L: the nodes of depth L eliminate themselves, their parents (depth L - 1) assemble.  The update matrices travel through the
// scratch (H itself is in registers by then and its staging area is free), two entries per 16-byte access, [pair][lane][2] - a store
// per pair and a load per pair and child; a lane gather (two ds_bpermute per double, ~48 cycles for a lone wavefront) costs eight
// times that.  All of it is one wavefront's own traffic: the LDS runs a wavefront's instructions in issue order, so a load behind a
// store needs the compiler to keep the order (wavefront-scope fences, rmx_lane_sync), not a wait for the store.
template <int L, int E>      // entry E of level L's update matrix: row E / (L + 1) + 1, column E % (L + 1) + 1 (the last one: the right-hand side)
__device__ __forceinline__ double& tree_entry(double (&G)[TREE_NR][TREE_RC + 1], const int up) {
    constexpr int k1 = E / (L + 1) + 1, kk = E % (L + 1);
    return G[k1 - up][kk < L ? kk + 1 - up : TREE_RC];
}
template <int L, int P>
__device__ __forceinline__ void tree_put(double (&G)[TREE_NR][TREE_RC + 1], double* __restrict__ X, const int lane) {
    if constexpr (P < L * (L + 1) / 2) {
        typedef double v2d __attribute__((ext_vector_type(2)));
        *reinterpret_cast<v2d*>(X + (P * 64 + lane) * 2) = v2d{tree_entry<L, 2 * P>(G, 0), tree_entry<L, 2 * P + 1>(G, 0)};
        tree_put<L, P + 1>(G, X, lane);
    }
}
// (the loads of a child's pairs in groups of TREE_GRP, every load of a group in flight before its first use: left alone, the compiler
// gives every load the same destination registers and waits for each - 112 LDS round trips per solve, 14 k cycles of its 16 k)
constexpr int TREE_GRP = 8;
typedef double tree_v2d __attribute__((ext_vector_type(2)));
template <int L, int P0, int I, int N>
__device__ __forceinline__ void tree_take_fma(double (&G)[TREE_NR][TREE_RC + 1], const tree_v2d (&v)[TREE_GRP], const double w) {
    if constexpr (I < N) {
        double& a = tree_entry<L, 2 * (P0 + I)>(G, 1);
        double& b = tree_entry<L, 2 * (P0 + I) + 1>(G, 1);
        a = fma(w, v[I][0], a);
        b = fma(w, v[I][1], b);
        tree_take_fma<L, P0, I + 1, N>(G, v, w);
    }
}
template <int L, int P0>
__device__ __forceinline__ void tree_take(double (&G)[TREE_NR][TREE_RC + 1], const double* __restrict__ Xc, const double w) {
    constexpr int NPAIR = L * (L + 1) / 2;
    if constexpr (P0 < NPAIR) {
        constexpr int N = NPAIR - P0 < TREE_GRP ? NPAIR - P0 : TREE_GRP;
        tree_v2d v[TREE_GRP];
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = *reinterpret_cast<const tree_v2d*>(Xc + (P0 + i) * 128);
        __builtin_amdgcn_sched_barrier(0);
        tree_take_fma<L, P0, 0, N>(G, v, w);
        __builtin_amdgcn_sched_barrier(0);
        tree_take<L, P0 + TREE_GRP>(G, Xc, w);
    }
}
template <int L>
__device__ __forceinline__ void tree_level(double (&G)[TREE_NR][TREE_RC + 1], const TreeLane& t, const int cmax, const int lane, double& rinv_own,
                                           const double (&lim)[TREE_DMAX + 1], GrowGuard& gm, PivGuard& pg, double* __restrict__ X) {
    const bool mine = t.depth == L;
    const double piv = mine ? G[0][0] : 1.0;
    const double rinv = recip(piv);
    pg.see(piv, rinv);
    rinv_own = mine ? rinv : rinv_own;
    double l[L + 1];
#pragma unroll
    for (int k1 = 1; k1 <= L; ++k1) {
        l[k1] = mine ? G[k1][0] * rinv : 0.0;
        // l'^2 = l^2 u_kk / d_anc <= LU_GROWTH_MAX^2 with l^2 u_kk = a l (see lu_solve_neg_diag): compared per ancestor against its
        // own bound, folded into one running maximum of the ratios' high words by scaling with 1 / lim
        gm.see((G[k1][0] * l[k1]) * lim[k1]);
    }
#pragma unroll
    for (int k1 = 1; k1 <= L; ++k1) {
#pragma unroll
        for (int k2 = 1; k2 <= L; ++k2) G[k1][k2] = fma(-l[k1], G[0][k2], G[k1][k2]);
        G[k1][TREE_RC] = fma(-l[k1], G[0][TREE_RC], G[k1][TREE_RC]);
    }
    rmx_lane_sync();            // (behind the previous level's loads - or the reads of H)
    tree_put<L, 0>(G, X, lane);
    rmx_lane_sync();
    // the parents take their children's update matrices, one level up
    const bool par = t.depth == L - 1;
#pragma unroll
    for (int c = 0; c < TREE_CMAX; ++c) {
        if (c < cmax) {
            const bool has = par && t.child[c] >= 0;
            tree_take<L, 0>(G, X + 2 * (has ? t.child[c] : lane), has ? 1.0 : 0.0);
        }
    }
}
// level L of the way down: every finished node's solution lies in the scratch ([lane]); the nodes of depth L read their ancestors'
template <int L>
__device__ __forceinline__ void tree_back(const double (&G)[TREE_NR][TREE_RC + 1], const TreeLane& t, const int lane, const double rinv_own,
                                          double& x, double* __restrict__ X) {
    const bool mine = t.depth == L;
    double v[L + 1];
#pragma unroll
    for (int k = 1; k <= L; ++k) v[k] = X[mine ? t.up[k] : lane];
    __builtin_amdgcn_sched_barrier(0);      // (every load in flight before the first use: see tree_take)
    double r = G[0][TREE_RC];
#pragma unroll
    for (int k = 1; k <= L; ++k) r = fma(-G[0][k], v[k], r);
    if (mine) x = r * rinv_own;
    rmx_lane_sync();
    if (mine) X[lane] = x;
    rmx_lane_sync();
}
// The Hessian stage's operands and the right-hand side are in place (eval_hess, the tree_dmax > 0 exit): sAcc = [H64_OP_ROWS + 1][H64_OP_STRIDE]
__device__ __forceinline__ double tree_solve64(const DevModel& M, const int lane, double* sAcc, bool& ok) {
    const double* sH = sAcc;                             // (H is read once, below; afterwards the scratch carries the update matrices)
    const TreeLane t = tree_lane(M, lane);
    const int dmax = M.tree_dmax, cmax = M.tree_cmax;
    const bool act = t.depth >= 0;
    double G[TREE_NR][TREE_RC + 1];
    double lim[TREE_DMAX + 1];
#pragma unroll
    for (int a = 0; a < TREE_NR; ++a)
#pragma unroll
        for (int b = 0; b <= TREE_RC; ++b) G[a][b] = 0.0;
    {
        // The entries of H this node's frontal matrix starts from, straight from the Hessian stage's operands ([k][node] rows of the
        // scratch, eval_hess): with a an ancestor of i,  H(i, a) = RL_i . CL_a  (12 terms: the `lo` product of the row of i with the
        // column of a) and  H(a, i) = RU_a . CU_i  (6 terms: the `up` product of the row of a with the column of i); the diagonal
        // travels as its own row.  One group of 19 loads per ancestor, all in flight before their first use.
        constexpr int ST = H64_OP_STRIDE;
        double rl[12], cu[6];
#pragma unroll
        for (int c = 0; c < 12; ++c) rl[c] = sH[(H64_R_RL + c) * ST + lane];
#pragma unroll
        for (int c = 0; c < 6; ++c) cu[c] = sH[(H64_R_CU + c) * ST + lane];
        const double hd0 = sH[H64_R_HD * ST + lane], rhs = sH[H64_OP_ROWS * ST + lane];
        __builtin_amdgcn_sched_barrier(0);
        G[0][0] = act ? hd0 : 1.0;
        G[0][TREE_RC] = act ? rhs : 0.0;
        lim[0] = 0.0;
#pragma unroll
        for (int k = 1; k <= TREE_DMAX; ++k) {
            if (k <= dmax) {
                const bool on = t.up[k] >= 0;
                const int aa = on ? t.up[k] : lane;
                double cl[12], ru[6];
#pragma unroll
                for (int c = 0; c < 12; ++c) cl[c] = sH[(H64_R_CL + c) * ST + aa];
#pragma unroll
                for (int c = 0; c < 6; ++c) ru[c] = sH[(H64_R_RU + c) * ST + aa];
                const double hda = sH[H64_R_HD * ST + aa];
                __builtin_amdgcn_sched_barrier(0);
                double lo = rl[0] * cl[0];
#pragma unroll
                for (int c = 1; c < 12; ++c) lo = fma(rl[c], cl[c], lo);
                double up = ru[0] * cu[0];
#pragma unroll
                for (int c = 1; c < 6; ++c) up = fma(ru[c], cu[c], up);
                G[0][k] = on ? lo : 0.0;                                    // H(i, ancestor)
                G[k][0] = on ? up : 0.0;                                    // H(ancestor, i)
                lim[k] = on ? recip((LU_GROWTH_MAX * LU_GROWTH_MAX) * hda) : 0.0;
            } else {
                lim[k] = 0.0;
            }
        }
    }
    GrowGuard gm;
    PivGuard pg;
    double rinv_own = 1.0;
    if (dmax >= 7) tree_level<7>(G, t, cmax, lane, rinv_own, lim, gm, pg, sAcc);
    if (dmax >= 6) tree_level<6>(G, t, cmax, lane, rinv_own, lim, gm, pg, sAcc);
    if (dmax >= 5) tree_level<5>(G, t, cmax, lane, rinv_own, lim, gm, pg, sAcc);
    if (dmax >= 4) tree_level<4>(G, t, cmax, lane, rinv_own, lim, gm, pg, sAcc);
    if (dmax >= 3) tree_level<3>(G, t, cmax, lane, rinv_own, lim, gm, pg, sAcc);
    if (dmax >= 2) tree_level<2>(G, t, cmax, lane, rinv_own, lim, gm, pg, sAcc);
    tree_level<1>(G, t, cmax, lane, rinv_own, lim, gm, pg, sAcc);
    static_assert(TREE_DMAX == 7, "tree_solve64: one call per level");
    // the root
    double x = 0.0;
    {
        const bool root = t.depth == 0;
        const double piv = root ? G[0][0] : 1.0;
        const double rinv = recip(piv);
        pg.see(piv, rinv);
        rinv_own = root ? rinv : rinv_own;
        x = root ? G[0][TREE_RC] * rinv : 0.0;
        rmx_lane_sync();        // (behind level 1's loads)
        sAcc[lane] = x;
        rmx_lane_sync();
    }
    tree_back<1>(G, t, lane, rinv_own, x, sAcc);
    if (dmax >= 2) tree_back<2>(G, t, lane, rinv_own, x, sAcc);
    if (dmax >= 3) tree_back<3>(G, t, lane, rinv_own, x, sAcc);
    if (dmax >= 4) tree_back<4>(G, t, lane, rinv_own, x, sAcc);
    if (dmax >= 5) tree_back<5>(G, t, lane, rinv_own, x, sAcc);
    if (dmax >= 6) tree_back<6>(G, t, lane, rinv_own, x, sAcc);
    if (dmax >= 7) tree_back<7>(G, t, lane, rinv_own, x, sAcc);
    // growth: the running maximum holds a_ik l / (64 d_anc), to stay at or below 1; pivots: every reciprocal positive and finite
    // (pivots are per lane here - every node has its own -, so both halves of the verdict are taken over the wavefront)
    ok = !__any(act && (gm.bad(1.0) || !pg.ok()));
    return act ? x : 0.0;
}

#if RMX_W2
// The solve as both waves of a workgroup call it (wave 0 from the Newton loop, wave 1 from w2_helper).
struct W2Lu {
    double dx;
    int ok;
};
#ifdef RMX_W2_LU_SHARED      // measurement aid: ONE out-of-line copy of the solve for both waves (416 B of callee-saved registers in scratch: 5 % slower)
__attribute__((noinline)) __device__ W2Lu w2_lu_call() {
#else
__device__ __forceinline__ W2Lu w2_lu_call() {
#endif
    extern __shared__ __attribute__((aligned(16))) double smem[];
    bool ok;
    const double dx = lu_solve_neg_diag64_staged(64, (int)(threadIdx.x & 63u), smem, ok);
    return W2Lu{dx, ok ? 1 : 0};
}
#endif

// the staged solve of the one-wave kernels: along the tree where the model has one to offer (DevModel::tree_dmax), else block columns
__device__ __forceinline__ double solve64_staged(const DevModel& M, const int lane, double* sAcc, bool& ok) {
    if (M.tree_dmax > 0) {
        const double dx = tree_solve64(M, lane, sAcc, ok);
        RMX_SYNC();                 // sAcc goes back to the front, whose subtree scan relies on a zero row n
        if (lane < ACC_STRIDE) sAcc[M.n * ACC_STRIDE + lane] = 0.0;
        RMX_SYNC();
        return dx;
    }
    return lu_solve_neg_diag64_staged(M.n, lane, sAcc, ok);
}

// the row-per-lane form of the interface (callers whose Hessian stage leaves H in registers)
__device__ __forceinline__ double lu_solve_neg_diag64(const int n, const int lane, double* sAcc, const double (&Hrow)[64], const double g,
                                                      bool& ok) {
    typedef double v2d __attribute__((ext_vector_type(2)));
    {
        v2d* w = reinterpret_cast<v2d*>(sAcc + lane * H64_STRIDE);
#pragma unroll
        for (int c = 0; c < 32; ++c) w[c] = v2d{Hrow[2 * c], Hrow[2 * c + 1]};
        sAcc[lane * H64_STRIDE + 64] = -g;
    }
    RMX_SYNC();
    return lu_solve_neg_diag64_staged(n, lane, sAcc, ok);
}

// BATCHED: pivot-row broadcasts in batches ahead of their FMAs (see lu_solve_neg_diag).  The Euler and adjoint kernels use
// it (-31 % on the solve); inside the Newton loop of the step kernels the same change costs the guarded path 2 % through
// register allocation, so the rare fallback there keeps the plain order.
template <int NP, bool BATCHED = false>
__device__ __forceinline__ double lu_solve_neg(const int n, const int lane, double (&Hrow)[NP], const double g) {
    // Rows/columns >= n are the identity (eval_node pads them), so all NP steps run unguarded: straight-line code lets
    // the scheduler overlap the pivot search of step k+1 with the trailing updates of step k.
    (void)n;
    double b = -g;
    int pivstep = (lane < NP) ? -1 : (NP + 1);   // -1: not yet used as a pivot row
    double rinv_own = 0.0;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        // every lane inverts its own candidate while the search runs (off the critical path)
        const double rinv_mine = recip(Hrow[k]);
        // pivot search: max |H(a,k)| over unused rows on the FULL double, the lowest row among equals: LAPACK's first maximum (idamax in
        // dgetf2 behind MATLAB's mldivide, driverRedMaxBDF1.m:117), the rule the large-tree kernels (rmx_big.hip) apply as well.  Three
        // integer reductions: the high words of |H| (order-preserving for non-negative doubles), then the low words of the lanes that
        // tie there, then the lane.  (Up to round 3 one reduction compared the top 26 bits only: rows closer than 2^-14 could be taken
        // in another order than LAPACK's.)
        const bool cand = pivstep < 0;
        const unsigned hiw = (unsigned)__double2hiint(Hrow[k]) & 0x7fffffffu, low = (unsigned)__double2loint(Hrow[k]);
        const unsigned k1 = cand ? (hiw | 0x80000000u) : 0u;
        const unsigned m1 = (NP <= 32) ? wave_umax32(k1) : wave_umax(k1);
        const bool t1 = cand && k1 == m1;
        const unsigned k2 = t1 ? low : 0u;
        const unsigned m2 = (NP <= 32) ? wave_umax32(k2) : wave_umax(k2);
        unsigned key = (t1 && low == m2) ? (64u - (unsigned)lane) : 0u;      // lanes 0..63 -> 64..1: the lowest lane has the largest key
        key = (NP <= 32) ? wave_umax32(key) : wave_umax(key);
        const int pl = 64 - (int)key;
        const double rinv = readlane_d(rinv_mine, pl);
        const bool elim = pivstep < 0 && lane != pl;
        if (lane == pl) {
            pivstep = k;
            rinv_own = rinv;
        }
        const double l = elim ? Hrow[k] * rinv : 0.0;
        // l == 0 on rows not being eliminated; the pivot row is broadcast in batches ahead of the FMAs (see lu_solve_neg_diag)
        constexpr int BT = NP > 32 ? 4 : LU_BATCH;
        if constexpr (!BATCHED) {
#pragma unroll
            for (int c = k + 1; c < NP; ++c) Hrow[c] -= l * readlane_d(Hrow[c], pl);
        } else
#pragma unroll
        for (int c0 = k + 1; c0 < NP; c0 += BT) {
            double pv[LU_BATCH];
#pragma unroll
            for (int i = 0; i < BT; ++i) pv[i] = (c0 + i < NP) ? readlane_d(Hrow[c0 + i < NP ? c0 + i : NP - 1], pl) : 0.0;
            if constexpr (BT == LU_BATCH) lu_pin(pv);
            else asm volatile("" : "+s"(pv[0]), "+s"(pv[1]), "+s"(pv[2]), "+s"(pv[3]));
#pragma unroll
            for (int i = 0; i < BT; ++i)
                if (c0 + i < NP) Hrow[c0 + i] -= l * pv[i];
        }
        b -= l * readlane_d(b, pl);
    }
    // back substitution on the implicitly permuted upper triangle
    double dx = 0.0;
#pragma unroll
    for (int k = NP - 1; k >= 0; --k) {
        const unsigned long long mk = __ballot(pivstep == k);
        const int pl = __builtin_amdgcn_readfirstlane((int)__ffsll((long long)mk) - 1);
        const double xk = readlane_d(b * rinv_own, pl);
        if (lane == k) dx = xk;
        if (pivstep < k) b -= Hrow[k] * xk;
    }
    return dx;
}

// ----------------------------------------------------------------------------- Newton
//
// Per-trajectory state of the linear-solve policy (lu_mode 0).  The Newton loop exists in two instantiations: the fast one
// (diagonal pivots under the growth guard, a tripped solve is redone with partial pivoting) and a pivot-only one.  The
// choice is made per STEP, outside the loop: putting the choice inside the loop cost 40 % on the 32-chain (9.9 -> 13.9 ms
// per 100 steps) through worse register allocation of the hot loop, whatever the policy did.  A trajectory whose solves
// tripped the guard PIV_STREAK times in a row (scenes whose diagonal ordering never passes, e.g. trees with prismatic
// joints) runs its next `len` steps pivot-only, doubling up to PIV_HOLD_MAX while the retries keep failing; isolated
// trips (the 32-chain) just pay for the redone solve.
constexpr int PIV_STREAK = 3, PIV_HOLD_MIN = 8, PIV_HOLD_MAX = 128;
struct PivotPolicy {
    int hold = 0;     // steps left that use the pivot-only Newton
    int len = 0;      // current hold length (0: not backing off)
    int streak = 0;   // consecutive tripped solves
};
__device__ __forceinline__ void pivot_policy_update(PivotPolicy& piv) {   // after a step of the fast Newton
    if (piv.streak >= PIV_STREAK) {
        piv.len = piv.len == 0 ? PIV_HOLD_MIN : (piv.len < PIV_HOLD_MAX ? 2 * piv.len : piv.len);
        piv.hold = piv.len;
        piv.streak = 0;
    } else if (piv.streak == 0) {
        piv.len = 0;
    }
}
//
// newton (driverRedMaxBDF1.m:94-157): damped Newton, backtracking on 0.5|g|^2 with strict decrease,
// at most iterLsMax halvings (the last trial is kept), stop on |g|<tol, iter>=iterMax or |dx|>dxMax.
// The (g,H) evaluation at the top of iteration k+1 is the Hessian stage applied to the state of the line-search
// evaluation that accepted x_{k+1} (same x, same arithmetic, so the same g the reference recomputes).
// LEAN (contact-capable kernels, see newton_node): the evaluations carry no contact terms but test whether any cuboid of the tree
// comes near the ground; the first one that does ends the solve with status bit 64 and the caller redoes it with CT = true.
constexpr int ST_LEFT_LEAN = 64;
constexpr int ST_LS_CUT = 128;     // RMX_ST_LS_CUT
constexpr int ST_PARK = 256;       // internal, like ST_LEFT_LEAN: the solve gave its rollout up to the cooperative launch
constexpr int ST_COOP_FAULT = 512; // RMX_ST_COOP_FAULT (include/redmax_hip.h): a member of a cooperative group waited in vain, or a parked rollout was never picked up; the rollout is invalid
//
// The cooperative line search (BASELINE.json configs[4]; DESIGN.md section 4 "park and relaunch").  At a stick / slip kink of the ground
// contact the reference's backtracking (driverRedMaxBDF2.m newton, the same as driverRedMaxBDF1.m:123-141) runs out its 20 trials - or
// accepts 2^-13 of the step - on every one of the 320 iterations of a step; its trial points alpha = 1, 1/2, ... are tested IN ORDER
// but their residuals are independent.  A rollout that meets such a step is parked (DevOpts::parkHalv) and finished by a GROUP of
// COOP_G wavefronts (one workgroup each, anywhere on the GPU) that all run the SAME Newton iteration redundantly - same code, same
// inputs, hence bitwise the same x0, dx, f0 in every member, no state to hand over - and differ in one thing only: in a line search
// member m evaluates trials 2 + 2 m and 3 + 2 m (eval_front_dual: two points per evaluation), i.e. all of trials 2 .. 21 at once.  What
// the members exchange is the four decision bits of their two points (stalled a / b, f < f0 a / b) in ONE 32-bit word per member and
// line search, tagged with the group's running count of line searches: a word is complete in itself, so relaxed agent-scope atomics
// carry it with no fence, and a reader simply polls until the tag is the current one.  Every member then walks the bits in the
// reference's order and reaches the reference's decision: identical iterates, iteration and halving counts, one evaluation latency
// per line search instead of ten.
#ifndef RMX_COOP_G
#define RMX_COOP_G 10
#endif
constexpr int COOP_G = RMX_COOP_G;                       // members of a group: trials 2 .. 2 COOP_G + 1 in one evaluation (iterLsMax = 20)
constexpr int COOP_WORDS = 32;                   // exchange words per group: 2 x COOP_G decision words (even / odd exchanges: a member may
                                                 // post exchange r + 1 while a slower one still reads r), [2 COOP_G] the group's abort flag
constexpr int COOP_REC = 40;                     // doubles per group the winner of a line search publishes (rmx_ct32.h CoopPub)
#ifndef RMX_COOP_SLEEP_X
#define RMX_COOP_SLEEP_X 8          // poll interval of the decision-word exchange, in 64-clock units (build variants)
#endif
#ifndef RMX_COOP_SLEEP_C
#define RMX_COOP_SLEEP_C 4          // poll interval of the wait for the winner's record
#endif
struct CoopCtx {
    unsigned* words = nullptr;                   // this group's COOP_WORDS exchange words (global memory)
    int member = 0;
    unsigned round = 0;                          // decision-word exchanges this group has gone through (the tag of the next one)
    unsigned rseq = 0;                           // records the group has passed on (rmx_ct32.h CoopPub: the tag of the next one)
    bool wide = false;                           // the line search in progress / the next one takes the whole group (rmx_ct32.h newton_pair)
    unsigned long long ticks = 5000000000ull;    // the longest wait for the group, in s_memtime ticks (DevOpts::coopTicks: ~2 s on this device)
#ifdef RMX_TICK_PHASE                            // measurement build (rmx_ct32.h RMX_PH_BEGIN): ticks of one phase of newton_pair
    unsigned long long phase = 0;
#endif
};
__device__ __forceinline__ unsigned coop_load(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void coop_store(unsigned* p, const unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// Post this member's decision bits for exchange `round` and collect the group's: lane j < COOP_G comes back with member j's word.
// false: some member did not answer within ~2 s of shader clock (or the group's abort flag was up): the caller gives the rollout up.
// (the exchange in two halves: a member may work between posting its word and collecting the group's - rmx_ct32.h newton_pair runs
// the Hessian stage of its own candidate trial there)
__device__ __forceinline__ void coop_post(CoopCtx& cx, const int lane, const unsigned bits) {
    ++cx.round;
    unsigned* const slot = cx.words + (cx.round & 1u) * COOP_G;
    if (lane == 0) coop_store(slot + cx.member, (cx.round << 4) | bits);
}
// iterLs / iterLsMax: the walk over the trials stops at the first member whose word ends the search - a stalled trial, an accepted one, or
// the one that is the last the reference allows - so the gather returns as soon as the words of the members up to and including that
// one are in (words of later members come back as "not arrived": the walk never reads them).
__device__ __forceinline__ bool coop_gather(CoopCtx& cx, const int lane, unsigned& word, const int iterLs, const int iterLsMax,
                                            const bool all = false) {
    const unsigned tag = cx.round << 4;
    unsigned* const slot = cx.words + (cx.round & 1u) * COOP_G;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    bool ok = true;
    word = tag;
    while (true) {
        unsigned w = tag;
        if (lane < COOP_G) w = coop_load(slot + lane);
        else if (lane == COOP_G) w = coop_load(cx.words + 2 * COOP_G) ? 0u : tag;    // the abort flag reads as a word that never arrives ...
        const bool mine = (w & ~15u) == tag;
        const bool ends = !all && lane < COOP_G && mine && ((w & 15u) != 0u || iterLs + 2 * lane + 1 >= iterLsMax);
        const unsigned long long arrived = __ballot(mine), ending = __ballot(ends);
        // every member up to the first one whose word ends the search has answered (lanes >= COOP_G always count as answered)
        const unsigned long long upto = ending ? ((ending & (0ull - ending)) << 1) - 1ull : ~0ull;
        if ((arrived & upto) == upto) {
            word = mine ? w : tag;
            break;
        }
        const bool aborted = __any(lane == COOP_G && !mine);                          // ... and ends the wait at once
        if (aborted || __builtin_amdgcn_s_memtime() - t0 > cx.ticks) {
            if (lane == 0) coop_store(cx.words + 2 * COOP_G, 1u);
            ok = false;
            break;
        }
        __builtin_amdgcn_s_sleep(RMX_COOP_SLEEP_X);
    }
    return ok;
}
__device__ __forceinline__ bool coop_exchange(CoopCtx& cx, const int lane, const unsigned bits, unsigned& word) {
    coop_post(cx, lane, bits);
    return coop_gather(cx, lane, word, 0, 0, true);      // (every word: this form is a plain all-to-all exchange)
}
#ifndef RMX_DUAL_LS
#define RMX_DUAL_LS 1              // two line-search points per evaluation (eval_front_dual); 0: one (build variants, measurements)
#endif
template <int NP, bool PIVOT_ONLY, bool CT = false, bool LEAN = false, bool COOP = false>
__device__ __forceinline__ double newton_impl(const DevModel& M, const DevOpts& o, double* sAcc, double* sCol, const int lane,
                                              double x, const double qA, const double qB, const double eta, NodeOut& last,
                                              int& iters, int& halvings, int& status, PivotPolicy& piv, double& xlo, CoopCtx& cx) {
    (void)sCol;
    (void)cx;
    const int halv_in = halvings;
    double Hrow[NP];
    FrontState fs;
    NodeOut e;
    // The iterate is carried as an unevaluated sum x + lo, |lo| <= ulp(x)/2 (o.comp = 1; 0 keeps lo = 0: plain doubles, the
    // reference's arithmetic).  g depends on x at the resolution of one ulp only where x enters LINEARLY with large coefficients:
    // v = x - qB (times the mass matrix) and qdot = (x - qA)/eta (times D); the geometry cannot resolve a fraction of an ulp of x
    // (sin, cos are rounded to their own ulp).  For the 32-link chain |M| ulp(x) ~ 1e-9 = the reference's tol: on the lattice of
    // doubles |g| < tol is reachable only at lucky points, and Newton finds them only if something dithers its update - in
    // MATLAB / the literal oracle their own evaluation noise does (0.1 % of steps fail), the world-frame evaluation has a smoother
    // error and sticks on 13 % of the steps (tools/reference_tol_stats.py).  With lo in v and qdot the
    // Newton correction is never lost to the rounding of x and the iteration converges quadratically to the evaluation noise
    // (~1e-10).  x is what a step stores (already rounded to nearest: two_sum), lo is dropped there.
    double lo = 0.0;
    eval_front<NP, true, false, CT, LEAN>(M, sAcc, lane, x, (x - qA) / eta, x - qB, eta, e, fs);
    if (LEAN && fs.touched) {
        status |= ST_LEFT_LEAN;
        return x;
    }
    int iter = 1, lsfail = 0;
    double gcarry = -1.0;
    while (true) {
        const double hdiag = eval_hess<NP, false, CT, PIVOT_ONLY>(M, lane, fs, Hrow, nullptr, sAcc, e.g);
        const NodeOut e0 = e;
        last = e;
        ++iters;
        double dx;
        if (PIVOT_ONLY) {
            dx = lu_solve_neg<NP>(M.n, lane, Hrow, e.g);
        } else {
            bool lu_ok;
            if constexpr (NP == 32 && LU_SPLIT32) dx = lu_solve_neg_diag32(M.n, lane, sAcc, e.g, lu_ok);
            else if constexpr (NP == 64 && LU_SPLIT64) {
                // the matrix-core Hessian stage (evaluations without contact terms: eval_hess compiles it for !CT only) has left H and
                // -g in the scratch; the v_readlane stage of the kernels with the contact terms hands the rows over in registers,
                // whether or not a corner touches the ground at this iterate
                if constexpr (HESS_MFMA64 && !CT) dx = solve64_staged(M, lane, sAcc, lu_ok);
                else dx = lu_solve_neg_diag64(M.n, lane, sAcc, Hrow, e.g, lu_ok);
            }
            else dx = lu_solve_neg_diag<NP>(lane, Hrow, e.g, hdiag, lu_ok);
            if (lu_ok) {
                piv.streak = 0;
            } else {             // growth guard tripped: redo this solve with partial pivoting
                ++piv.streak;
                status |= 16;
                // H was destroyed in place.  The front is re-evaluated too (same x, same arithmetic) so that its state does
                // not have to stay live in registers across the fast-path LU for the sake of this rare branch.
                NodeOut e2;
                eval_front<NP, true, false, CT, LEAN>(M, sAcc, lane, x, ((x - qA) + lo) / eta, (x - qB) + lo, eta, e2, fs);
                eval_hess<NP, false, CT>(M, lane, fs, Hrow, nullptr, sAcc);
                dx = lu_solve_neg<NP>(M.n, lane, Hrow, e.g);
            }
        }
        const double dxn2 = wave_sum(dx * dx);
        if (!(dxn2 == dxn2)) {   // NaN: give up on this trajectory instead of spinning to iterMax
            status |= 4;
            break;
        }
        if (sqrt(dxn2) > o.dxMax) {
            status |= 1;         // "Newton diverged" (:118-121): x is left at the last iterate
            break;
        }
        double alpha = 1.0;
        // |g(x0)|^2: from the second iteration on it is the |g|^2 the previous line search ended with (same vector, same
        // deterministic reduction), so it is carried over instead of being reduced again
        const double g0n2 = gcarry >= 0.0 ? gcarry : wave_sum(e.g * e.g);
        const double f0 = 0.5 * g0n2;
        const double x0 = x, lo0 = lo;
        int iterLs = 1;
        double gn2 = g0n2;
        bool stalled = false;
        // chains of at most 32 nodes in the kernels with the contact terms: from the second trial on the line search evaluates its
        // points two at a time (eval_front_dual) - same points, same order of decisions
        constexpr bool DUAL_LS = CT && !LEAN && NP == 32 && RMX_DUAL_LS;
        while (true) {
            if constexpr (DUAL_LS && COOP) {
                static_assert(!COOP || (CT && !LEAN && NP == 32), "the cooperative line search belongs to the 32-lane kernels with the contact terms");
                if (iterLs >= 2 && M.is_chain) {
                    // member m: trial iterLs + 2 m at alpha 4^-m (lanes 0..31) and trial iterLs + 2 m + 1 at half of that (lanes 32..63)
                    const double x0d = dup_lo(x0), lo0d = dup_lo(lo0), dxd = dup_lo(dx), qAd = dup_lo(qA), qBd = dup_lo(qB);
                    const bool hiH = lane >= 32;
                    const double am = ldexp(alpha, -2 * cx.member);           // (alpha is a power of two: exact)
                    const double al = hiH ? 0.5 * am : am;
                    double xl, lol;
                    two_sum(x0d, fma(al, dxd, lo0d), xl, lol);
                    lol *= o.comp;
                    const unsigned long long same = __ballot(xl == x0d && lol == lo0d);
                    const bool stall_a = (unsigned)same == 0xffffffffu, stall_b = (unsigned)(same >> 32) == 0xffffffffu;
                    unsigned bits = (stall_a ? 1u : 0u) | (stall_b ? 2u : 0u);
                    if (!stall_a) {       // (a stalled point a: b and every later member's points are stalled too, nobody looks at their f)
                        const double gd = eval_front_dual<CT>(RMX_CONSTS(sAcc, M.n, NP), Grav3{M.grav[0], M.grav[1], M.grav[2]}, lane, xl,
                                                              ((xl - qAd) + lol) / eta, (xl - qBd) + lol, eta, dup_lo(fs.tau_add));
                        double ga2, gb2;
                        wave_sum_dual(gd * gd, ga2, gb2);
                        bits |= (0.5 * ga2 < f0 ? 4u : 0u) | (0.5 * gb2 < f0 ? 8u : 0u);
                    }
                    unsigned word;
                    if (!coop_exchange(cx, lane, bits, word)) {
                        status |= 4 | ST_COOP_FAULT;
                        xlo = lo0;
                        return x0;
                    }
                    // the reference's walk over the trials, in order (see the two-point path below)
                    int take = -1;                         // the trial (counted from iterLs) that ends the search
#pragma unroll 1
                    for (int m = 0; m < COOP_G && take < 0 && !stalled; ++m) {
                        const unsigned w = (unsigned)__builtin_amdgcn_readlane((int)word, m);
                        if (w & 1u) { stalled = true; break; }
                        if ((w & 4u) || iterLs + 2 * m >= o.iterLsMax) { take = 2 * m; break; }
                        if (w & 2u) { stalled = true; break; }
                        if ((w & 8u) || iterLs + 2 * m + 1 >= o.iterLsMax) { take = 2 * m + 1; break; }
                    }
                    if (stalled) {
                        iterLs = o.iterLsMax;
                        e = e0;
                        x = x0;
                        lo = lo0;
                        break;
                    }
                    if (take >= 0) {
                        iterLs += take;
                        two_sum(x0, fma(ldexp(alpha, -take), dx, lo0), x, lo);       // what the dual layout holds for that point, lane = node
                        lo *= o.comp;
                        x = hiH ? x0 : x;
                        lo = hiH ? lo0 : lo;
                        eval_front<NP, true, false, CT, LEAN>(M, sAcc, lane, x, ((x - qA) + lo) / eta, (x - qB) + lo, eta, e, fs);
                        gn2 = wave_sum(e.g * e.g);
                        break;
                    }
                    alpha = ldexp(alpha, -2 * COOP_G);
                    iterLs += 2 * COOP_G;
                    continue;
                }
            } else if constexpr (DUAL_LS) {
                if (iterLs >= 2 && M.is_chain) {
                    // trial iterLs at alpha (lanes 0..31) and trial iterLs + 1 at alpha / 2 (lanes 32..63), node = lane & 31
                    const double x0d = dup_lo(x0), lo0d = dup_lo(lo0), dxd = dup_lo(dx), qAd = dup_lo(qA), qBd = dup_lo(qB);
                    const bool hiH = lane >= 32;
                    const double al = hiH ? 0.5 * alpha : alpha;
                    double xl, lol;
                    two_sum(x0d, fma(al, dxd, lo0d), xl, lol);
                    lol *= o.comp;
                    const unsigned long long same = __ballot(xl == x0d && lol == lo0d);
                    const bool stall_a = (unsigned)same == 0xffffffffu, stall_b = (unsigned)(same >> 32) == 0xffffffffu;
                    int take = -1;                         // 0: trial point a ends the search, 1: b
                    if (stall_a) {                         // (see the one-point path below)
                        stalled = true;
                        iterLs = o.iterLsMax;
                        e = e0;
                        x = x0;
                        lo = lo0;
                        break;
                    }
                    const double gd = eval_front_dual<CT>(RMX_CONSTS(sAcc, M.n, NP), Grav3{M.grav[0], M.grav[1], M.grav[2]}, lane, xl,
                                                          ((xl - qAd) + lol) / eta, (xl - qBd) + lol, eta, dup_lo(fs.tau_add));
                    double ga2, gb2;
                    wave_sum_dual(gd * gd, ga2, gb2);
                    if (0.5 * ga2 < f0 || iterLs >= o.iterLsMax) {
                        take = 0;
                    } else {
                        ++iterLs;                          // trial point b
                        if (stall_b) {
                            stalled = true;
                            iterLs = o.iterLsMax;
                            e = e0;
                            x = x0;
                            lo = lo0;
                            break;
                        }
                        if (0.5 * gb2 < f0 || iterLs >= o.iterLsMax) take = 1;
                    }
                    if (take >= 0) {
                        // the point that ends the search: the full front there (the state the Hessian stage needs; its |g|^2 is the
                        // one carried on, as in the one-point path)
                        const double xs = take ? take_hi(xl) : xl, ls = take ? take_hi(lol) : lol;
                        x = hiH ? x0 : xs;
                        lo = hiH ? lo0 : ls;
                        eval_front<NP, true, false, CT, LEAN>(M, sAcc, lane, x, ((x - qA) + lo) / eta, (x - qB) + lo, eta, e, fs);
                        gn2 = wave_sum(e.g * e.g);
                        break;
                    }
                    alpha *= 0.25;
                    ++iterLs;
                    continue;
                }
            }
            two_sum(x0, fma(alpha, dx, lo0), x, lo);       // x + lo = x0 + (lo0 + alpha dx)
            lo *= o.comp;
            if (__all(x == x0 && lo == lo0)) {
                // alpha*dx no longer changes the iterate in any DOF: this and every further halving re-evaluates g at x0
                // bit-for-bit, so f == f0 is never a strict decrease, the reference runs out its iterLsMax trials and
                // keeps x0 (:132-138).  Same outcome, without the evaluations.
                stalled = true;
                iterLs = o.iterLsMax;
                e = e0;              // the evaluation at x0
                break;
            }
            eval_front<NP, true, false, CT, LEAN>(M, sAcc, lane, x, ((x - qA) + lo) / eta, (x - qB) + lo, eta, e, fs);
            if (LEAN && fs.touched) {
                status |= ST_LEFT_LEAN;
                return x;
            }
            gn2 = wave_sum(e.g * e.g);
            if (0.5 * gn2 < f0) break;
            if (iterLs >= o.iterLsMax) break;
            alpha *= 0.5;
            ++iterLs;
        }
        last = e;
        halvings += iterLs - 1;
        if constexpr (DUAL_LS && !COOP) {
            // (see CoopCtx) a solve whose line searches keep running out their trials: hand the rollout over at the start of this step
            if (o.parkHalv > 0 && halvings - halv_in > o.parkHalv && M.is_chain) {
                status |= ST_PARK;
                return x;
            }
        }
        if (stalled) {
            // g is g(x0) again.  If it is not below tol the next Newton iteration is this one repeated exactly, and so
            // on until iter >= iterMax ("Newton did not converge", :150-153) with x unchanged: report that now.
            if (!(sqrt(g0n2) < o.tol)) status |= 2 | 8;
            break;
        }
        gcarry = gn2;
        if (sqrt(gn2) < o.tol) break;
        if (iter >= o.iterMax) {
            status |= 2;         // "Newton did not converge" (:150-153)
            break;
        }
        // rmx_opts.ls_fail_limit (off by default): a line search that ran out its trials without a decrease leaves x0 + 2^-19 dx;
        // the reference goes on to iterMax, failing the same way every time (a non-smooth point of g: stick/slip, touch-down)
        lsfail += (0.5 * gn2 < f0) ? 0 : 1;          // counted over the step, not consecutive: failed and barely successful ones alternate
        if (o.lsFailLimit > 0 && lsfail >= o.lsFailLimit) {
            status |= 2 | ST_LS_CUT;
            break;
        }
        ++iter;
    }
    xlo = lo;
    return x;
}

// The same Newton with the loop ROTATED so that the kernel holds ONE call site of the front per instantiation: the evaluation at the
// top of the loop body is the first evaluation of the solve on entry and a line-search trial point afterwards (`ls`, wave-uniform).
// In newton_impl the state of the front (FrontState, ~70 doubles per lane) has two producers - the evaluation before the loop and the
// one inside the line search - and the compiler reconciles their register assignments with blocks of pure moves on the loop's edges
// (~180 v_mov / v_accvgpr per Newton iteration of the 32-link chain kernel, 5 % of its issue slots).  Same decisions in the same
// order, same arithmetic: the results are bit-identical.  Plain kernels only (the contact-capable ones keep newton_impl with its
// two-point line search and the lean exit).
#ifndef RMX_NEWTON_ROT
#define RMX_NEWTON_ROT 1
#endif
template <int NP, bool PIVOT_ONLY>
__device__ __forceinline__ double newton_rot(const DevModel& M, const DevOpts& o, double* sAcc, double* sCol, const int lane,
                                             double x, const double qA, const double qB, const double eta, NodeOut& last,
                                             int& iters, int& halvings, int& status, PivotPolicy& piv, double& xlo) {
    (void)sCol;
    double Hrow[NP];
    FrontState fs;
    NodeOut e, e0;
    double lo = 0.0, dx = 0.0, alpha = 1.0, f0 = 0.0, g0n2 = 0.0, x0 = x, lo0 = 0.0;
    int iter = 1, lsfail = 0, iterLs = 1;
    bool ls = false;
    e0.g = e0.eT = e0.eV = 0.0;
    while (true) {
        eval_front<NP, true, false, false, false>(M, sAcc, lane, x, ((x - qA) + lo) / eta, (x - qB) + lo, eta, e, fs);
        const double gn2 = wave_sum_np<NP>(e.g * e.g);
        if (ls) {                                        // this was a trial point of the line search (:124-138)
            if (!(0.5 * gn2 < f0) && iterLs < o.iterLsMax) {
                alpha *= 0.5;
                ++iterLs;
                two_sum(x0, fma(alpha, dx, lo0), x, lo);
                lo *= o.comp;
                if (__all(x == x0 && lo == lo0)) {        // see newton_impl: every further halving re-evaluates g(x0)
                    last = e0;
                    halvings += o.iterLsMax - 1;
                    if (!(sqrt(g0n2) < o.tol)) status |= 2 | 8;
                    break;
                }
                continue;
            }
            last = e;
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
        const double hdiag = eval_hess<NP, false, false, PIVOT_ONLY>(M, lane, fs, Hrow, nullptr, sAcc, e.g);
        e0 = e;
        last = e;
        ++iters;
        if (PIVOT_ONLY) {
            dx = lu_solve_neg<NP>(M.n, lane, Hrow, e.g);
        } else {
            bool lu_ok;
            if constexpr (NP == 32 && LU_SPLIT32) dx = lu_solve_neg_diag32(M.n, lane, sAcc, e.g, lu_ok);
            else if constexpr (NP == 64 && LU_SPLIT64) {
#if RMX_W2
                if (M.tree_dmax > 0) {   // (the helper wave stays out of this one: w2_helper)
                    dx = tree_solve64(M, lane, sAcc, lu_ok);
                } else {
                    const W2Lu r = w2_lu_call();
                    dx = r.dx;
                    lu_ok = r.ok != 0;
                }
                RMX_SYNC();             // sAcc goes back to the front, whose subtree scan relies on a zero row n
                if (lane < ACC_STRIDE) sAcc[M.n * ACC_STRIDE + lane] = 0.0;
                RMX_SYNC();
#else
                if constexpr (HESS_MFMA64) dx = solve64_staged(M, lane, sAcc, lu_ok);
                else dx = lu_solve_neg_diag64(M.n, lane, sAcc, Hrow, e.g, lu_ok);
#endif
            }
            else dx = lu_solve_neg_diag<NP>(lane, Hrow, e.g, hdiag, lu_ok);
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
            last = e0;
            halvings += o.iterLsMax - 1;
            if (!(sqrt(g0n2) < o.tol)) status |= 2 | 8;
            break;
        }
        ls = true;
    }
    xlo = lo;
    return x;
}

template <int NP, bool CT, bool LEAN, bool COOP = false>
__device__ __forceinline__ double newton_policy(const DevModel& M, const DevOpts& o, double* sAcc, double* sCol, const int lane,
                                                double x, const double qA, const double qB, const double eta, NodeOut& last,
                                                int& iters, int& halvings, int& status, PivotPolicy& piv, double& xlo, CoopCtx& cx) {
    constexpr bool ROT = RMX_NEWTON_ROT && !CT && !LEAN;
    if (o.lu_mode != 0 || piv.hold > 0) {     // wave-uniform
        if (piv.hold > 0) --piv.hold;
        if constexpr (ROT) return newton_rot<NP, true>(M, o, sAcc, sCol, lane, x, qA, qB, eta, last, iters, halvings, status, piv, xlo);
        else return newton_impl<NP, true, CT, LEAN, COOP>(M, o, sAcc, sCol, lane, x, qA, qB, eta, last, iters, halvings, status, piv, xlo, cx);
    }
    double r;
    if constexpr (ROT) r = newton_rot<NP, false>(M, o, sAcc, sCol, lane, x, qA, qB, eta, last, iters, halvings, status, piv, xlo);
    else r = newton_impl<NP, false, CT, LEAN, COOP>(M, o, sAcc, sCol, lane, x, qA, qB, eta, last, iters, halvings, status, piv, xlo, cx);
    pivot_policy_update(piv);
    return r;
}

// One implicit solve of a step.  LEAN (the first of the two launches of a contact-capable step, rmx_kernels.hip): the plain
// evaluation plus a test that every cuboid of the tree is clear of the ground, under which the contact terms vanish
// identically.  The first evaluation that fails the test ends the solve: status bit ST_LEFT_LEAN comes back set, nothing else
// is touched, and the caller parks the trajectory at the start of this step for the launch with the contact terms.
template <int NP, bool CT = false, bool LEAN = false, bool COOP = false>
__device__ __forceinline__ double newton_node(const DevModel& M, const DevOpts& o, double* sAcc, double* sCol, const int lane,
                                              double x, const double qA, const double qB, const double eta, NodeOut& last,
                                              int& iters, int& halvings, int& status, PivotPolicy& piv, double& xlo, CoopCtx& cx) {
    xlo = 0.0;
    if constexpr (!LEAN) {
        return newton_policy<NP, CT, false, COOP>(M, o, sAcc, sCol, lane, x, qA, qB, eta, last, iters, halvings, status, piv, xlo, cx);
    } else {
        int it2 = 0, hv2 = 0, st2 = 0;
        PivotPolicy pv2 = piv;
        NodeOut l2;
        const double r = newton_policy<NP, false, true>(M, o, sAcc, sCol, lane, x, qA, qB, eta, l2, it2, hv2, st2, pv2, xlo, cx);
        if (st2 & ST_LEFT_LEAN) {
            status |= ST_LEFT_LEAN;
            return x;
        }
        iters += it2;
        halvings += hv2;
        status |= st2;
        piv = pv2;
        last = l2;
        return r;
    }
}

}  // namespace rmx
