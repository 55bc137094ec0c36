// smart_fast_entry.h -- the kernels of SMART_MATH_FAST, one __global__ per arithmetic variant.
//
// Round 1 had ONE kernel that held every variant behind a wave-uniform switch; it was allocated the registers of
// its hungriest branch (189 VGPRs, 209 spilled SGPRs: two wavefronts per SIMD for everything).  Each variant is now
// its own kernel with its own register allocation (profiles/r02_kernel_resources.txt), launched over the same grid:
// a workgroup classifies its 64 parameter rows (wave_class) and returns at once unless they belong to its kernel.
// The host launches only the kernels a call can need (report type, gap, outputs; and, with a plan from
// smart_plan_ensemble, only the classes actually present) -- concurrently, on forked streams, when there are several.
//
//   kernel                       rows / forcing it takes                                         body
//   smart_fast_intervals_exits   class 0, summary, gap >= 2, piecewise-constant forcing          interval engine, early exits
//   smart_fast_intervals         same, for launches with <= 2 blocks of 64 samples per SIMD    interval engine, straight-line
//   smart_fast_intervals_states  same, final state vector asked for                              interval engine, SPLIT
//   smart_fast_steps             class 0, summary, gap >= 2, forcing varying inside the interval step loop, merged
//   smart_fast_steps_states      same, final state vector asked for                              step loop, SPLIT
//   smart_fast_plain             class 0, raw reports or gap 1                                   step loop
//   smart_fast_stiff             class 1: some k * 3600 < dt (clamps, 95 % rule reachable)        step loop, STIFF
//   smart_fast_guard             class 2: S outside [0, 0.5], C < 0 or Z <= 0                     step loop, GUARD
//   smart_fast_illcond           class 3: some dt / (k * 3600) > 2                                literal model
#pragma once

#include "smart_fast_model.h"

namespace smart {

enum FastKernel : int {
    kIntervalsExits = 0,
    kIntervals,
    kIntervalsStates,
    kSteps,
    kStepsStates,
    kPlain,
    kStiff,
    kGuard,
    kIllCond,
    kNumFastKernels
};

// ticket counters of the two families of time-sliced kernels (workspace header, claim_work)
constexpr int kTicketIntervals = 0, kTicketSteps = 1;

// Do this block's 64 rows belong to the kernel of class CLS?  A block whose class has no kernel in this call (the
// caller's plan is stale) is reported through the status word by whichever kernel meets it first.
template <int CLS>
__device__ __forceinline__ bool block_is_mine(const KArgs &a, const Work &w)
{
    const int cls = wave_class(a, w.block, w.c);
    if (cls == CLS)
        return true;
    if (w.seg == 0 && !((a.class_mask >> cls) & 1))
        raise_status(a, kStatusStalePlan);
    return false;
}

// Class-0 summary runs: is this catchment's forcing the kind this kernel takes?  Answered by smart_forcing_scan
// before the launch (a.not_pc); without a workspace every wavefront scans the forcing itself.
template <bool PIECEWISE>
__device__ __forceinline__ bool forcing_is_mine(const KArgs &a, const double2 *__restrict__ forcing, const Work &w)
{
    const bool pc = a.not_pc ? a.not_pc[w.c] == 0 : forcing_is_piecewise_constant(forcing + w.c * a.T, a.T, a.gap);
    if (pc == PIECEWISE)
        return true;
    if (w.seg == 0 && !((a.pc_mask >> (pc ? 0 : 1)) & 1))
        raise_status(a, kStatusStalePlan);
    return false;
}

template <class Model, bool PIECEWISE>
__device__ __forceinline__ void merged_kernel(const KArgs &a, const double2 *__restrict__ forcing,
                                              const double *__restrict__ obs, const double *__restrict__ ws)
{
    const Work w = claim_work(a, PIECEWISE ? kTicketIntervals : kTicketSteps);
    if (!block_is_mine<0>(a, w) || !forcing_is_mine<PIECEWISE>(a, forcing, w))
        return;
    run_ensemble_merged<Model, PIECEWISE>(a, forcing, obs, ws, w.block, w.c, w.seg);
}

#define SMART_FAST_KERNEL(name)                                                                                        \
    __global__ __launch_bounds__(kWave) void name(KArgs a, const double2 *__restrict__ forcing,                        \
                                                  const double *__restrict__ obs, const double *__restrict__ ws)

// host stub of a kernel (for hipLaunchKernel / the occupancy query); each translation unit answers for its own
const void *fast_kernel_intervals(FastKernel k);
const void *fast_kernel_steps(FastKernel k);
const void *fast_kernel_guarded(FastKernel k);

} // namespace smart
