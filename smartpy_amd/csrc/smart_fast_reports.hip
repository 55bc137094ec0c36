// smart_fast_reports.hip -- the regular (class 0) rows under the reports that are not interval means: report='raw' (the
// outflow of the last step of each interval, structure.py:192-195) and a report every step (gap 1, :190 with a mean over
// one value).  Round 3 ran both through smart_fast_plain, the general step loop, unsliced (24.8 / 43.7 ms at 1e5 samples
// x hourly 10 years next to the summary run's 10.2); here they get what the summary kernels have -- the asm arms, the
// interval engine, the time-sliced launch (run_ensemble_merged<..., REPORT> in smart_device.h).  Kernels of their own,
// in a translation unit of their own, so that the summary kernels keep their code and registers.
// See smart_fast_entry.h for the family.
#include "smart_fast_entry.h"

namespace smart {

SMART_FAST_KERNEL(smart_fast_steps_raw) { merged_kernel<FastModel<false, false, true>, kForcingVarying, kReportLast>(a, forcing, obs, ws); }

SMART_FAST_KERNEL(smart_fast_intervals_raw)
{
    merged_kernel<FastModel<false, false, true, false>, kForcingIntervals, kReportLast>(a, forcing, obs, ws);
}

SMART_FAST_KERNEL(smart_fast_steps_every) { merged_kernel<FastModel<false, false, true>, kForcingVarying, kReportEvery>(a, forcing, obs, ws); }

const void *fast_kernel_reports(FastKernel k)
{
    switch (k) {
    case kStepsRaw:
        return reinterpret_cast<const void *>(&smart_fast_steps_raw);
    case kIntervalsRaw:
        return reinterpret_cast<const void *>(&smart_fast_intervals_raw);
    case kStepsEvery:
        return reinterpret_cast<const void *>(&smart_fast_steps_every);
    default:
        return nullptr;
    }
}

} // namespace smart
