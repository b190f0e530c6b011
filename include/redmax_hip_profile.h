/*
 * redmax_hip_profile.h -- measurement hooks of libredmax_hip.so, kept OUT of the host-facing ABI (include/redmax_hip.h).
 *
 * Nothing here replaces a line of the reference: these entries exist for bench.py, tools/ and the 'timing' / 'ticks' commands of the
 * MEX gateway.  A host that only simulates (driverRedMaxBDF1.m's simLoop) binds redmax_hip.h alone.  Same
 * conventions (extern "C", 0 / negative RMX_E_* return codes, host pointers copied out).
 */
#ifndef REDMAX_HIP_PROFILE_H
#define REDMAX_HIP_PROFILE_H

#include "redmax_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Timing hook for benchmarks: milliseconds spent in the kernels of the last rmx_step_* call, measured
 * with hipEvents on the batch's own stream. */
double rmx_last_step_ms(const rmx_batch* b);
/* The step kernel the last rmx_step_* call of this batch launched, as a short label ("k_step_bdf1_pair32", "k_step_bdf1<32,fullchain>",
 * "k_step_bdf1<64,w2>", "k_ground32", "k_big_step", ...): which of the size / batch / environment dependent variants the library chose.
 * A static string (never null; "" before the first step call).  What bench.py labels its roofline object with. */
const char* rmx_last_step_kernel(const rmx_batch* b);
/* Profiling hook: mean shader-clock cycles per wavefront of {residual evaluation, residual+Hessian evaluation,
 * LU solve, the two norm reductions} of one Newton iteration at the current state (reps repetitions per trajectory)
 * in cycles16[0..3]; cycles16[4..15] split the residual+Hessian evaluation into its 12 stages (rmx_device.h RMX_STAMP). */
int rmx_profile_phases(rmx_batch* b, int reps, double h, double* cycles16);
/* Per-rollout share of the last rmx_step_bdf1 / bdf2 / history / bdf1_async launch: ticks[batch] = shader-clock ticks (s_memtime) each
 * rollout's wavefront spent inside the kernel(s) of that call.  All rollouts of a batch run concurrently (one wavefront each) and the
 * launch ends with the slowest: the distribution (median, 99th percentile, maximum) says how much of the launch time is its tail. */
int rmx_step_ticks(rmx_batch* b, unsigned long long* ticks);

#ifdef __cplusplus
}
#endif
#endif
