// smart_fast_intervals.hip -- the interval engine (run_ensemble_merged<..., kForcingIntervals>): summary reports over forcing that
// is constant within the report interval, the reference's own input format (daily values spread over the hours of the
// day, timeframe.py:167-186).  The headline kernels.  See smart_fast_entry.h for the family.
#include "smart_fast_entry.h"

namespace smart {

SMART_FAST_KERNEL(smart_fast_intervals_exits) { merged_kernel<FastModel<false, false, true, true>, kForcingIntervals>(a, forcing, obs, ws); }

SMART_FAST_KERNEL(smart_fast_intervals) { merged_kernel<FastModel<false, false, true, false>, kForcingIntervals>(a, forcing, obs, ws); }

SMART_FAST_KERNEL(smart_fast_intervals_states)
{
    merged_kernel<FastModel<false, false, true, true, true>, kForcingIntervals>(a, forcing, obs, ws);
}

const void *fast_kernel_intervals(FastKernel k)
{
    switch (k) {
    case kIntervalsExits:
        return reinterpret_cast<const void *>(&smart_fast_intervals_exits);
    case kIntervals:
        return reinterpret_cast<const void *>(&smart_fast_intervals);
    case kIntervalsStates:
        return reinterpret_cast<const void *>(&smart_fast_intervals_states);
    default:
        return nullptr;
    }
}

} // namespace smart
