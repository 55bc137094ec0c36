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

// the ill-conditioned rows (dt / RK > 2) and any row with a NaN or an infinite parameter (wave_class(): class 3) on
// the literal model -- which is why this translation unit is compiled WITHOUT -fno-honor-nans (build.py): what a NaN
// does in the reference's compares and branches is part of what these kernels reproduce, bit for bit with the literal
// kernel (tests/test_gpu_parity.py: ..._wild_parameters_...).
//
// TWO forms of the same arithmetic, the same bits (tests/test_gpu_parity.py: ..._two_forms_...), chosen per launch from
// the number of class-3 blocks the plan counted (smart_capi.hip: illcond_form(); round 6):
//
//   smart_fast_illcond        the LATENCY form (round 5).  One sample per DPP row, SIXTEEN workgroups per block of 64
//                             samples (grid.x = 16 ceil(N / 64), kIllCondWaves), four samples each.  A wavefront alone
//                             on its SIMD is as fast as its instruction count, and this form's step is a third of the
//                             other's -- but it advances 4 samples where the other advances 64: right while sixteen
//                             wavefronts per block still find SIMDs of their own (config 2: 19 blocks, 4.67 -> 1.52 ms).
//   smart_fast_illcond_lanes  the THROUGHPUT form (round 4's, LiteralModelT<true>).  One sample per lane, one workgroup
//                             per block.  ~490 instructions per step for 64 samples against ~176 for 4: from a few
//                             hundred class-3 blocks on (a daily ensemble of 1e5 samples and more) the row form queues
//                             for SIMDs round after round and this one does not.
SMART_FAST_KERNEL(smart_fast_illcond)
{
    Work w;
    w.block = (long)(blockIdx.x >> 4);
    w.c = (long)blockIdx.y;
    w.seg = (int)(blockIdx.x & 15u); // (0 for the block's first wavefront: the one that reports a stale plan)
    if (!block_is_mine<3>(a, w))
        return;
    if (w.block * kWave + w.seg * 4 >= a.N)
        return; // (the batch ends before this wavefront's four samples)
    run_ensemble<LiteralLanesModel, false>(a, forcing, obs, ws, nullptr, w.block, w.c, w.seg);
}

SMART_FAST_KERNEL(smart_fast_illcond_lanes) { guarded_kernel<3, LiteralModelT<true>>(a, forcing, obs, ws); }

const void *fast_kernel_guarded(FastKernel k)
{
    switch (k) {
    case kStiff:
        return reinterpret_cast<const void *>(&smart_fast_stiff);
    case kGuard:
        return reinterpret_cast<const void *>(&smart_fast_guard);
    case kIllCond:
        return reinterpret_cast<const void *>(&smart_fast_illcond);
    case kIllCondLanes:
        return reinterpret_cast<const void *>(&smart_fast_illcond_lanes);
    default:
        return nullptr;
    }
}

} // namespace smart
