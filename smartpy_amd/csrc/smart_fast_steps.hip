// smart_fast_steps.hip -- the step loops of the regular (class 0) rows: forcing that varies inside the report interval
// (genuinely sub-daily data: the asm arms of smart_fast_arms.h), raw reports, and report gaps of one step.  See
// smart_fast_entry.h for the family.
#include "smart_fast_entry.h"

namespace smart {

// SMART_STEPS_WAVES (A/B builds, round 6): ask hipcc for an allocation that lets that many wavefronts of smart_fast_steps
// share a SIMD (3: <= 168 VGPRs where it takes 170 by itself; profiles/r06_ab_steps_third_wave.txt has what that buys)
#ifdef SMART_STEPS_WAVES
#define SMART_STEPS_OCC __attribute__((amdgpu_waves_per_eu(SMART_STEPS_WAVES, SMART_STEPS_WAVES)))
#else
#define SMART_STEPS_OCC
#endif
SMART_STEPS_OCC SMART_FAST_KERNEL(smart_fast_steps) { merged_kernel<FastModel<false, false, true>, kForcingVarying>(a, forcing, obs, ws); }

SMART_FAST_KERNEL(smart_fast_steps_states) { merged_kernel<FastModel<false, false, true, true, true>, kForcingVarying>(a, forcing, obs, ws); }

// raw reports / gap 1: the merged form when only discharge and the groundwater ratio are asked for, the five
// reservoirs carried separately when the caller wants the final state vector
SMART_FAST_KERNEL(smart_fast_plain)
{
    const Work w = claim_work(a, 0);
    if (!block_is_mine<0>(a, w))
        return;
    if (a.final_vars == nullptr)
        run_ensemble<FastModel<false, false, true>, false>(a, forcing, obs, ws, nullptr, w.block, w.c);
    else
        run_ensemble<FastModel<false, false>, false>(a, forcing, obs, ws, nullptr, w.block, w.c);
}

const void *fast_kernel_steps(FastKernel k)
{
    switch (k) {
    case kSteps:
        return reinterpret_cast<const void *>(&smart_fast_steps);
    case kStepsStates:
        return reinterpret_cast<const void *>(&smart_fast_steps_states);
    case kPlain:
        return reinterpret_cast<const void *>(&smart_fast_plain);
    default:
        return nullptr;
    }
}

} // namespace smart
