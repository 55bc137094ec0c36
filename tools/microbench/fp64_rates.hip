// fp64_rates.hip -- what does one SIMD of gfx950 sustain on the fp64 vector instructions the SMART step uses?
// Each wave runs ITER iterations of an unrolled block of 16 instructions of one kind, either as 8 independent
// chains (throughput) or as one dependent chain (latency).  Grids of 1, 2 and 4 waves per SIMD; optionally only
// the lower 32 (or 16) lanes active, to see whether the hardware skips inactive halves of a wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITER 20000

#define OP8(INS)                                                                                                     \
    asm volatile(INS " %0, %8, %9, %0\n" INS " %1, %8, %9, %1\n" INS " %2, %8, %9, %2\n" INS " %3, %8, %9, %3\n"      \
                 INS " %4, %8, %9, %4\n" INS " %5, %8, %9, %5\n" INS " %6, %8, %9, %6\n" INS " %7, %8, %9, %7\n"      \
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)                     \
                 : "v"(a), "v"(b))
#define OP8_2(INS)                                                                                                   \
    asm volatile(INS " %0, %8, %0\n" INS " %1, %8, %1\n" INS " %2, %8, %2\n" INS " %3, %8, %3\n"                      \
                 INS " %4, %8, %4\n" INS " %5, %8, %5\n" INS " %6, %8, %6\n" INS " %7, %8, %7\n"                      \
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)                     \
                 : "v"(a))
#define DEP8(INS)                                                                                                    \
    asm volatile(INS " %0, %1, %2, %0\n" INS " %0, %1, %2, %0\n" INS " %0, %1, %2, %0\n" INS " %0, %1, %2, %0\n"      \
                 INS " %0, %1, %2, %0\n" INS " %0, %1, %2, %0\n" INS " %0, %1, %2, %0\n" INS " %0, %1, %2, %0\n"      \
                 : "+v"(x0) : "v"(a), "v"(b))
#define DEP8_2(INS)                                                                                                  \
    asm volatile(INS " %0, %1, %0\n" INS " %0, %1, %0\n" INS " %0, %1, %0\n" INS " %0, %1, %0\n"                      \
                 INS " %0, %1, %0\n" INS " %0, %1, %0\n" INS " %0, %1, %0\n" INS " %0, %1, %0\n"                      \
                 : "+v"(x0) : "v"(a))

template <int KIND, int LANES>
__global__ __launch_bounds__(64) void k(double *out, const double *in)
{
    if ((int)threadIdx.x >= LANES) return;
    double a = in[0], b = in[1];
    double x0 = in[2] + threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < ITER; ++i) {
        if (KIND == 0) { OP8("v_fma_f64"); OP8("v_fma_f64"); }
        if (KIND == 1) { DEP8("v_fma_f64"); DEP8("v_fma_f64"); }
        if (KIND == 2) { OP8_2("v_add_f64"); OP8_2("v_add_f64"); }
        if (KIND == 3) { OP8_2("v_mul_f64"); OP8_2("v_mul_f64"); }
        if (KIND == 4) { OP8_2("v_min_f64"); OP8_2("v_min_f64"); }
        if (KIND == 5) { DEP8_2("v_add_f64"); DEP8_2("v_add_f64"); }
        if (KIND == 7) { DEP8_2("v_min_f64"); DEP8_2("v_min_f64"); }
    }
    out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

template <int KIND, int LANES>
static void run(const char *name, double *out, double *in)
{
    for (int wps : {1, 2, 4, 8}) {
        const int grid = 1024 * wps;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL((k<KIND, LANES>), dim3(grid), dim3(64), 0, 0, out, in);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<KIND, LANES>), dim3(grid), dim3(64), 0, 0, out, in);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double insts = (double)ITER * 16;                  // per wave
        const double ns_per_inst_per_simd = ms * 1e6 / (insts * wps);
        printf("%-22s lanes=%2d waves/SIMD=%d  %8.3f ms  %6.3f ns per wave-instruction per SIMD (= %5.2f cyc @2.4GHz)\n", name, LANES,
               wps, ms, ns_per_inst_per_simd, ns_per_inst_per_simd * 2.4);
    }
}

int main()
{
    double *out, *in;
    hipMalloc(&out, 8192 * 64 * sizeof(double));
    hipMalloc(&in, 64);
    double h[3] = {1.0000001, 1e-9, 0.5};
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    run<0, 64>("v_fma_f64 8 chains", out, in);
    run<1, 64>("v_fma_f64 dependent", out, in);
    run<2, 64>("v_add_f64 8 chains", out, in);
    run<5, 64>("v_add_f64 dependent", out, in);
    run<3, 64>("v_mul_f64 8 chains", out, in);
    run<4, 64>("v_min_f64 8 chains", out, in);
    run<7, 64>("v_min_f64 dependent", out, in);
    run<0, 32>("v_fma_f64 8 chains", out, in);
    run<0, 16>("v_fma_f64 8 chains", out, in);
    run<1, 32>("v_fma_f64 dependent", out, in);
    return 0;
}
