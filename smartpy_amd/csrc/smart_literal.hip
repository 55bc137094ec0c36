// smart_literal.hip -- SMART_MATH_LITERAL: kernels around the literal model (smart_literal_model.h).
//
// Compiled with -ffp-contract=off like the model itself, so that the shared loop skeleton (report means,
// groundwater sums) rounds every operation separately too: discharge, groundwater ratio and final state of this
// kernel are bit-identical to the CPU oracle configured with the product chain for s'**i
// (tests/test_gpu_parity.py), which pins every branch and every operation order of the GPU path.
//
// This mode is the parity anchor, not the fast path: ~39 (wet) / 26 (dry) fp64 divisions per step.
#include "smart_literal_lanes.h"

namespace smart {

__global__ __launch_bounds__(kWave) void smart_ensemble_literal(KArgs a, const double2 *__restrict__ forcing,
                                                                const double *__restrict__ obs,
                                                                const double *__restrict__ ws)
{
    extern __shared__ double lds[];
    if (a.np_mean)
        run_ensemble<LiteralModel, true>(a, forcing, obs, ws, lds, (long)blockIdx.x, (long)blockIdx.y);
    else
        run_ensemble<LiteralModel, false>(a, forcing, obs, ws, lds, (long)blockIdx.x, (long)blockIdx.y);
}

// The kernel behind smartcpp.allsteps (smart_allsteps_hip: one sample, a latency chain of one wavefront): the same loop
// on LiteralLanesModel (smart_literal_lanes.h) -- the sample spread over the sixteen lanes of a DPP row, six layers and
// five reservoirs one instruction each, the reference's sums as v_fmac_f64_dpp chains in the reference's order, the
// divisions by per-sample constants through cached reciprocals + correction step: the division's bits, a third of
// the instructions per step.  Grid: sixteen wavefronts per block of 64 samples, four samples each (one: a single
// wavefront).  The ensemble's literal mode above keeps the true divisions, one sample per lane: it is the anchor this
// form is compared with (tests/test_gpu_parity.py, tests/test_gpu_api.py).
__global__ __launch_bounds__(kWave) void smart_ensemble_literal_rows(KArgs a, const double2 *__restrict__ forcing,
                                                                     const double *__restrict__ obs,
                                                                     const double *__restrict__ ws)
{
    extern __shared__ double lds[];
    const long block = (long)(blockIdx.x >> 4);
    const int sub = (int)(blockIdx.x & 15u);
    if (block * kWave + sub * 4 >= a.N)
        return;
    if (a.np_mean)
        run_ensemble<LiteralLanesModel, true>(a, forcing, obs, ws, lds, block, (long)blockIdx.y, sub);
    else
        run_ensemble<LiteralLanesModel, false>(a, forcing, obs, ws, lds, block, (long)blockIdx.y, sub);
}

// smartcpp.onestep stand-in: n independent single steps (structure.py:200-264)
__global__ __launch_bounds__(kWave) void smart_onestep_literal(long n, const double *in, double *out)
{
    const long i = (long)blockIdx.x * kWave + threadIdx.x;
    if (i >= n)
        return;
    const double *x = in + i * 26;
    double p[10], st[12];
#pragma unroll
    for (int k = 0; k < 10; ++k)
        p[k] = x[4 + k];
#pragma unroll
    for (int k = 0; k < 12; ++k)
        st[k] = x[14 + k];
    LiteralModel m;
    m.setup(x[0], x[1], p);
    m.set_states(st);
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    m.step(x[2], x[3], 0.0, s0, s1, s2);
    double v[19];
    m.get_vars(v, nullptr);
#pragma unroll
    for (int k = 0; k < 19; ++k)
        out[i * 19 + k] = v[k];
}

// run_one_step_river stand-in: n independent river steps; in = (dt, q_in, rk [hours], V), out = (q_out, V')
__global__ __launch_bounds__(kWave) void smart_river_literal(long n, const double *in, double *out)
{
    const long i = (long)blockIdx.x * kWave + threadIdx.x;
    if (i >= n)
        return;
    const double *x = in + i * 4;
    double v = x[3];
    const double q = LiteralModel::river(x[0], x[1], x[2] * 3600.0, v); // :482
    out[i * 2] = q;
    out[i * 2 + 1] = v;
}

void launch_river(long n, const double *in, double *out, hipStream_t s)
{
    hipLaunchKernelGGL(smart_river_literal, dim3((unsigned)((n + kWave - 1) / kWave)), dim3(kWave), 0, s, n, in, out);
}

void launch_literal(const KArgs &a, dim3 grid, size_t lds_bytes, hipStream_t s, bool rows)
{
    if (rows)
        hipLaunchKernelGGL(smart_ensemble_literal_rows, dim3(grid.x * kIllCondWaves, grid.y), dim3(kWave), lds_bytes, s,
                           a, reinterpret_cast<const double2 *>(a.forcing), a.obs, a.ws);
    else
        hipLaunchKernelGGL(smart_ensemble_literal, grid, dim3(kWave), lds_bytes, s, a,
                           reinterpret_cast<const double2 *>(a.forcing), a.obs, a.ws);
}

void launch_onestep(long n, const double *in, double *out, hipStream_t s)
{
    hipLaunchKernelGGL(smart_onestep_literal, dim3((unsigned)((n + kWave - 1) / kWave)), dim3(kWave), 0, s, n, in, out);
}

} // namespace smart
