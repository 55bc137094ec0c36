// lone.hip -- what does ONE wavefront that has its SIMD to itself issue per cycle, and what decides it?
// (round 4: the time loops at 1e5 samples run at 1.5 wavefronts per SIMD; a 4-byte shift of a loop moved it by 4.7 %.)
// One loop of 64 fp64 instructions per variant, written as asm so that the compiler has no say:
//   chains   number of independent accumulators the 64 instructions rotate over (1 = every instruction waits for
//            the one before it)
//   shift    0: the run of 64-bit encodings starts on an 8-byte boundary; 1: at 4 mod 8
//   salu     a scalar instruction after every `salu` vector ones (0 = none)
// Timed in the wave itself: s_memtime (shader clock) and s_memrealtime (100 MHz) around the loop; the slowest wave of
// the grid is reported.  Grids of 1 and 2 waves per SIMD (1024 / 2048 one-wave blocks on an idle chip; placement is one
// per SIMD: profiles/r01_microbench_wave_placement.txt).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define FMA(d) "v_fma_f64 " d ", " d ", v[20:21], v[22:23]\n\t"
#define S_ "s_add_u32 s20, s20, 1\n\t"

// 16 instructions rotating over C chains (registers v[0:1] .. v[14:15])
#define R1 FMA("v[0:1]") FMA("v[0:1]") FMA("v[0:1]") FMA("v[0:1]")
#define R2 FMA("v[0:1]") FMA("v[2:3]") FMA("v[0:1]") FMA("v[2:3]")
#define R4 FMA("v[0:1]") FMA("v[2:3]") FMA("v[4:5]") FMA("v[6:7]")
#define R8A FMA("v[0:1]") FMA("v[2:3]") FMA("v[4:5]") FMA("v[6:7]")
#define R8B FMA("v[8:9]") FMA("v[10:11]") FMA("v[12:13]") FMA("v[14:15]")

#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v20","v21","v22","v23","s20","s21","scc"

template <int CHAINS, int SHIFT, int SALU>
__device__ __forceinline__ void body(int iters)
{
    // 64 vector instructions per trip
#define Q(X) X X X X
#define LOOP(BODY)                                                                                                  \
    asm volatile("s_mov_b32 s21, %0\n\t"                                                                            \
                 "s_mov_b32 s20, 0\n\t"                                                                             \
                 ".p2align 6\n\t" PAD "1:\n\t" BODY "s_sub_u32 s21, s21, 1\n\t"                                     \
                 "s_cmp_lg_u32 s21, 0\n\t"                                                                          \
                 "s_cbranch_scc1 1b\n\t" ::"s"(iters)                                                               \
                 : CLOB)
    if (SHIFT == 0) {
#define PAD ""
        if (SALU == 0) {
            if (CHAINS == 1) LOOP(Q(Q(R1)));
            if (CHAINS == 2) LOOP(Q(Q(R2)));
            if (CHAINS == 4) LOOP(Q(Q(R4)));
            if (CHAINS == 8) LOOP(Q(Q(R8A R8B)) );
        } else if (SALU == 4) {
            if (CHAINS == 1) LOOP(Q(Q(R1 S_)));
            if (CHAINS == 4) LOOP(Q(Q(R4 S_)));
        } else if (SALU == 8) {
            if (CHAINS == 1) LOOP(Q(Q(R1 R1 S_)) );
            if (CHAINS == 4) LOOP(Q(Q(R4 R4 S_)) );
        } else if (SALU == 44) {   // two scalar instructions after every 4: the parity stays
            if (CHAINS == 1) LOOP(Q(Q(R1 S_ S_)));
            if (CHAINS == 4) LOOP(Q(Q(R4 S_ S_)));
        }
#undef PAD
    } else {
#define PAD "s_nop 0\n\t"
        if (SALU == 0) {
            if (CHAINS == 1) LOOP(Q(Q(R1)));
            if (CHAINS == 2) LOOP(Q(Q(R2)));
            if (CHAINS == 4) LOOP(Q(Q(R4)));
            if (CHAINS == 8) LOOP(Q(Q(R8A R8B)) );
        }
#undef PAD
    }
}

template <int CHAINS, int SHIFT, int SALU>
__global__ __launch_bounds__(64) void k(unsigned long long *rec, const double *in, int iters)
{
    asm volatile("v_mov_b32 v20, %0\n\tv_mov_b32 v21, %1\n\tv_mov_b32 v22, %2\n\tv_mov_b32 v23, %3\n\t"
                 "v_mov_b32 v0, 0\n\tv_mov_b32 v1, 0\n\tv_mov_b32 v2, 0\n\tv_mov_b32 v3, 0\n\t"
                 "v_mov_b32 v4, 0\n\tv_mov_b32 v5, 0\n\tv_mov_b32 v6, 0\n\tv_mov_b32 v7, 0\n\t"
                 "v_mov_b32 v8, 0\n\tv_mov_b32 v9, 0\n\tv_mov_b32 v10, 0\n\tv_mov_b32 v11, 0\n\t"
                 "v_mov_b32 v12, 0\n\tv_mov_b32 v13, 0\n\tv_mov_b32 v14, 0\n\tv_mov_b32 v15, 0\n\t" ::"v"(
                     __double2loint(in[0])),
                 "v"(__double2hiint(in[0])), "v"(__double2loint(in[1])), "v"(__double2hiint(in[1]))
                 : CLOB);
    const unsigned long long c0 = __builtin_readcyclecounter(), t0 = __builtin_amdgcn_s_memrealtime();
    body<CHAINS, SHIFT, SALU>(iters);
    const unsigned long long c1 = __builtin_readcyclecounter(), t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        rec[blockIdx.x * 2 + 0] = c1 - c0;
        rec[blockIdx.x * 2 + 1] = t1 - t0;
    }
}

template <int CHAINS, int SHIFT, int SALU>
static void run(unsigned long long *rec, const double *in)
{
    const int iters = 4000;
    for (int wps : {1, 2, 3}) {
        const int grid = 1024 * wps;
        hipLaunchKernelGGL((k<CHAINS, SHIFT, SALU>), dim3(grid), dim3(64), 0, 0, rec, in, 10);
        hipDeviceSynchronize();
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<CHAINS, SHIFT, SALU>), dim3(grid), dim3(64), 0, 0, rec, in, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h((size_t)grid * 2);
        hipMemcpy(h.data(), rec, h.size() * 8, hipMemcpyDeviceToHost);
        unsigned long long cmax = 0, tmax = 0, cmin = ~0ull;
        for (int b = 0; b < grid; ++b) {
            cmax = std::max(cmax, h[b * 2]);
            cmin = std::min(cmin, h[b * 2]);
            tmax = std::max(tmax, h[b * 2 + 1]);
        }
        const double n = 64.0 * iters;
        printf("chains=%d start=%d mod 8 salu-every=%2d waves/SIMD=%d | launch %.3f ms | slowest wave %.3f ms = %.2f ns per "
               "vector instruction per wave, %.2f per SIMD slot (x2.4 GHz: %.2f cycles) | s_memtime ticks per instruction "
               "%.2f (fastest wave %.2f)\n",
               CHAINS, SHIFT * 4, SALU, wps, ms, tmax / 1e5, tmax * 10.0 / n, tmax * 10.0 / n / wps, tmax * 10.0 / n / wps * 2.4,
               cmax / n, cmin / n);
    }
}

int main()
{
    unsigned long long *rec;
    double *in;
    hipMalloc(&rec, 4096 * 2 * 8);
    hipMalloc(&in, 64);
    double h[2] = {1.0000001, 1e-9};
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    run<1, 0, 0>(rec, in);
    run<1, 1, 0>(rec, in);
    run<2, 0, 0>(rec, in);
    run<2, 1, 0>(rec, in);
    run<4, 0, 0>(rec, in);
    run<4, 1, 0>(rec, in);
    run<8, 0, 0>(rec, in);
    run<8, 1, 0>(rec, in);
    run<1, 0, 4>(rec, in);
    run<4, 0, 4>(rec, in);
    run<1, 0, 8>(rec, in);
    run<4, 0, 8>(rec, in);
    run<1, 0, 44>(rec, in);
    run<4, 0, 44>(rec, in);
    return 0;
}
