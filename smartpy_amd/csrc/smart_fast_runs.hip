// smart_fast_runs.hip -- the interval engine over RUNS of constant forcing shorter than the report interval
// (run_ensemble_merged<..., kForcingRuns>): 3-, 6- or 12-hourly data in an hourly run with daily reports, which the
// reference's own input pipeline produces (timeframe.py:167-186 spreads each value equally over the steps it covers).
// Every property the interval engine rests on holds per run -- fixed wet / dry side, additive evaporation, linear dry
// map -- so a run of k steps (k a divisor of the gap, found by smart_forcing_scan) is advanced at once and the
// report mean accumulates across the gap / k runs of an interval.  Kernels of their own so that the headline kernels
// (smart_fast_intervals.hip) keep their code and registers.  See smart_fast_entry.h for the family.
#include "smart_fast_entry.h"

namespace smart {

SMART_FAST_KERNEL(smart_fast_runs_exits) { merged_kernel<FastModel<false, false, true, true>, kForcingRuns>(a, forcing, obs, ws); }

SMART_FAST_KERNEL(smart_fast_runs) { merged_kernel<FastModel<false, false, true, false>, kForcingRuns>(a, forcing, obs, ws); }

SMART_FAST_KERNEL(smart_fast_runs_states)
{
    merged_kernel<FastModel<false, false, true, true, true>, kForcingRuns>(a, forcing, obs, ws);
}

const void *fast_kernel_runs(FastKernel k)
{
    switch (k) {
    case kRunsExits:
        return reinterpret_cast<const void *>(&smart_fast_runs_exits);
    case kRuns:
        return reinterpret_cast<const void *>(&smart_fast_runs);
    case kRunsStates:
        return reinterpret_cast<const void *>(&smart_fast_runs_states);
    default:
        return nullptr;
    }
}

} // namespace smart
