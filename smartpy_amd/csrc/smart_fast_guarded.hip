// smart_fast_guarded.hip -- the rows outside the regular class: reachable clamps and 95 % rule (STIFF), leak guards
// that can matter (GUARD), and the ill-conditioned rows that only the reference's own operation order reproduces
// (literal model).  Daily steps with the default parameter ranges put a third of the rows here; hourly steps none.
// See smart_fast_entry.h for the family.
#include "smart_fast_entry.h"
#include "smart_literal_lanes.h"

namespace smart {

template <int CLS, class Model>
__device__ __forceinline__ void guarded_kernel(const KArgs &a, const double2 *__restrict__ forcing,
                                               const double *__restrict__ obs, const double *__restrict__ ws)
{
    const Work w = claim_work(a, 0);
    if (!block_is_mine<CLS>(a, w))
        return;
    run_ensemble<Model, false>(a, forcing, obs, ws, nullptr, w.block, w.c);
}

SMART_FAST_KERNEL(smart_fast_stiff) { guarded_kernel<1, FastModel<true, false>>(a, forcing, obs, ws); }

SMART_FAST_KERNEL(smart_fast_guard) { guarded_kernel<2, FastModel<true, true>>(a, forcing, obs, ws); }

#ifndef SMART_ILLCOND_ROWS
#define SMART_ILLCOND_ROWS 1 // one sample per DPP row (smart_literal_lanes.h; 0: one per lane, round 4's form -- A/B builds)
#endif
// the ill-conditioned rows (dt / RK > 2) and any row with a NaN or an infinite parameter (wave_class(): class 3) on
// the literal model -- which is why this translation unit is compiled WITHOUT -fno-honor-nans (build.py): what a NaN
// does in the reference's compares and branches is part of what this kernel reproduces, bit for bit with the literal
// kernel (tests/test_gpu_parity.py: ..._wild_parameters_...).
// Round 5: launched over SIXTEEN workgroups per block of 64 samples (grid.x = 16 ceil(N / 64), kIllCondWaves), each
// taking four of the block's samples, one per DPP row: such rows are few (a tenth of a daily ensemble: 19 blocks of
// config 2), their wavefronts are alone on their SIMDs, and a lone wavefront's time is its instruction count
// (profiles/r05_microbench_lanes.txt: 4.67 -> 1.52 ms for config 2's 1,160 rows, every bit the same).
SMART_FAST_KERNEL(smart_fast_illcond)
{
#if SMART_ILLCOND_ROWS
    Work w;
    w.block = (long)(blockIdx.x >> 4);
    w.c = (long)blockIdx.y;
    w.seg = (int)(blockIdx.x & 15u); // (0 for the block's first wavefront: the one that reports a stale plan)
    if (!block_is_mine<3>(a, w))
        return;
    if (w.block * kWave + w.seg * 4 >= a.N)
        return; // (the batch ends before this wavefront's four samples)
    run_ensemble<LiteralLanesModel, false>(a, forcing, obs, ws, nullptr, w.block, w.c, w.seg);
#else
    guarded_kernel<3, LiteralModelT<true>>(a, forcing, obs, ws);
#endif
}

const void *fast_kernel_guarded(FastKernel k)
{
    switch (k) {
    case kStiff:
        return reinterpret_cast<const void *>(&smart_fast_stiff);
    case kGuard:
        return reinterpret_cast<const void *>(&smart_fast_guard);
    case kIllCond:
        return reinterpret_cast<const void *>(&smart_fast_illcond);
    default:
        return nullptr;
    }
}

} // namespace smart
