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
//   smart_fast_runs(_exits)      same as the two above for forcing constant over runs of k steps,  interval engine over runs
//   smart_fast_runs_states       k >= 2 a divisor of the gap (e.g. 6-hourly data, hourly steps)     ... SPLIT
//   smart_fast_steps             class 0, summary, gap >= 2, forcing varying inside the interval step loop, merged
//   smart_fast_steps_states      same, final state vector asked for                              step loop, SPLIT
//   smart_fast_steps_raw         class 0, raw reports, gap >= 2 (W, T multiples of it)           step loop, last-step routing
//   smart_fast_intervals_raw     same over piecewise-constant forcing                            interval engine, n - 1 + 1 steps
//   smart_fast_steps_every       class 0, a report every step (gap 1)                            single-step arms + report
//   smart_fast_plain             class 0, what is left: raw reports over a ragged time axis, raw   step loop (unsliced)
//                                or every-step reports with the final state vector
//   smart_fast_stiff             class 1: some k * 3600 < dt (clamps, 95 % rule reachable)        step loop, STIFF
//   smart_fast_guard             class 2: S outside [0, 0.5], C < 0 or Z <= 0                     step loop, GUARD
//   smart_fast_illcond           class 3: dt / (RK * 3600) > 2 (the river only), few blocks      literal model, one sample per
//                                                                                                 DPP row (latency form)
//   smart_fast_illcond_lanes     class 3, many blocks (round 6: chosen per launch from the load)  literal model, one sample per
//                                                                                                 lane, cached reciprocals
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
    kRunsExits,
    kRuns,
    kRunsStates,
    kStepsRaw,
    kIntervalsRaw,
    kStepsEvery,
    kIllCondLanes,
    kNumFastKernels
};

// ticket counters of the families of time-sliced kernels (workspace header, claim_work): at most one kernel of a
// family runs in a call
constexpr int kTicketIntervals = 0, kTicketSteps = 1, kTicketRuns = 2, kTicketStepsRaw = 3, kTicketIntervalsRaw = 4,
              kTicketEvery = 5;
static_assert(kHdrTicket + kTicketEvery < 8, "the plan word of smart_plan_ensemble sits at header int 8");

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
// before the launch (a.fflags); without a workspace every wavefront scans the forcing itself.
template <int FORCING>
__device__ __forceinline__ bool forcing_is_mine(const KArgs &a, int fflags, const Work &w)
{
    const int kind = forcing_kind(a, fflags);
    if (kind == FORCING)
        return true;
    // pc_mask: bit 0 the interval engine, bit 1 the step loop, bit 2 the run engine
    const int bit = kind == kForcingIntervals ? 0 : (kind == kForcingVarying ? 1 : 2);
    if (w.seg == 0 && !((a.pc_mask >> bit) & 1))
        raise_status(a, kStatusStalePlan);
    return false;
}

template <class Model, int FORCING, int REPORT = kReportMean>
__device__ __forceinline__ void merged_kernel(const KArgs &a, const double2 *__restrict__ forcing,
                                              const double *__restrict__ obs, const double *__restrict__ ws)
{
    constexpr int ticket = REPORT == kReportEvery  ? kTicketEvery
                           : REPORT == kReportLast ? (FORCING == kForcingIntervals ? kTicketIntervalsRaw : kTicketStepsRaw)
                           : FORCING == kForcingIntervals ? kTicketIntervals
                                                          : (FORCING == kForcingVarying ? kTicketSteps : kTicketRuns);
    const Work w = claim_work(a, ticket);
    if (!block_is_mine<0>(a, w))
        return;
    const int fflags = forcing_flags(a, forcing, w.c);
    if constexpr (REPORT == kReportLast && FORCING == kForcingVarying) {
        // the step loop of raw reports takes every forcing the interval engine does not (varying, and constant over runs
        // shorter than the report interval: there is no run engine for raw reports)
        if (forcing_kind(a, fflags) == kForcingIntervals) {
            if (w.seg == 0 && !(a.pc_mask & 1))
                raise_status(a, kStatusStalePlan);
            return;
        }
    } else if constexpr (REPORT != kReportEvery) { // (a report every step: one kernel whatever the forcing)
        if (!forcing_is_mine<FORCING>(a, fflags, w))
            return;
    }
    run_ensemble_merged<Model, FORCING, REPORT>(a, forcing, obs, ws, w.block, w.c, w.seg, fflags);
}

#define SMART_FAST_KERNEL(name)                                                                                        \
    __global__ __launch_bounds__(kWave) void name(KArgs a, const double2 *__restrict__ forcing,                        \
                                                  const double *__restrict__ obs, const double *__restrict__ ws)

// host stub of a kernel (for hipLaunchKernel / the occupancy query); each translation unit answers for its own
const void *fast_kernel_intervals(FastKernel k);
const void *fast_kernel_steps(FastKernel k);
const void *fast_kernel_guarded(FastKernel k);
const void *fast_kernel_runs(FastKernel k);
const void *fast_kernel_reports(FastKernel k);

} // namespace smart
