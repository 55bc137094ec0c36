// duo.hip -- would it pay to give a block of 64 samples TWO wavefronts, one for the soil recurrence (60 of a wet step's
// 73 vector instructions) and one for the routing recurrence (13), the first feeding the second three values per step
// through LDS?  At 1e5 samples a SIMD holds one or two wavefronts and issues a vector instruction every 4.7-5.6 cycles
// instead of every 4; the routing wave would fill the gaps.  Here: chains of dependent fp64 FMAs of the same lengths,
//   solo   64 threads:  73 FMAs per step
//   duo   128 threads:  wave 0: 60 FMAs + 3 ds_write_b64 per step, s_barrier every 4 steps
//                       wave 1: s_barrier, then 4 x (3 ds_read_b64, 13 FMAs)
// on grids of 1 and 2 blocks per SIMD (1,024 / 2,048 blocks; 256 CUs x 4 SIMDs).
#include <hip/hip_runtime.h>
#include <cstdio>

#define STEPS 40000

#define FMA8 "v_fma_f64 %0, %8, %9, %0\nv_fma_f64 %1, %8, %9, %1\nv_fma_f64 %2, %8, %9, %2\nv_fma_f64 %3, %8, %9, %3\n" \
             "v_fma_f64 %4, %8, %9, %4\nv_fma_f64 %5, %8, %9, %5\nv_fma_f64 %6, %8, %9, %6\nv_fma_f64 %7, %8, %9, %7\n"
#define FMA4 "v_fma_f64 %0, %8, %9, %0\nv_fma_f64 %1, %8, %9, %1\nv_fma_f64 %2, %8, %9, %2\nv_fma_f64 %3, %8, %9, %3\n"
#define FMA5 FMA4 "v_fma_f64 %4, %8, %9, %4\n"
#define FMA1 "v_fma_f64 %0, %8, %9, %0\n"
#define REGS : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b)

__global__ __launch_bounds__(64) void solo(double *out, const double *in)
{
    double a = in[0], b = in[1];
    double x0 = in[2] + threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < STEPS; ++i)
        asm volatile(FMA8 FMA8 FMA8 FMA8 FMA8 FMA8 FMA8 FMA8 FMA8 FMA1 REGS); // 73
    out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

__global__ __launch_bounds__(128) void duo(double *out, const double *in)
{
    __shared__ double buf[2][4][3][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double a = in[0], b = in[1];
    double x0 = in[2] + lane, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    if (wave == 0) {
        for (int g = 0; g < STEPS / 4; ++g) {
            double(*h)[3][64] = buf[g & 1];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                asm volatile(FMA8 FMA8 FMA8 FMA8 FMA8 FMA8 FMA8 FMA4 REGS); // 60
                h[j][0][lane] = x0;
                h[j][1][lane] = x1;
                h[j][2][lane] = x2;
            }
            __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0)
            __builtin_amdgcn_s_barrier();
        }
    } else {
        for (int g = 0; g < STEPS / 4; ++g) {
            double(*h)[3][64] = buf[g & 1];
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                x5 += h[j][0][lane];
                x6 += h[j][1][lane];
                x7 += h[j][2][lane];
                asm volatile(FMA8 FMA1 FMA1 REGS); // 10 + the 3 adds = 13
            }
        }
    }
    out[blockIdx.x * 128 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

int main()
{
    double *out, *in, h[3] = {0.999999, 1e-9, 1.0};
    hipMalloc(&out, 8192 * 128 * 8);
    hipMalloc(&in, 3 * 8);
    hipMemcpy(in, h, 24, hipMemcpyHostToDevice);
    for (int per : {1, 2, 3, 4}) {
        const int grid = 1024 * per;
        float ms[2];
        for (int which = 0; which < 2; ++which) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (which == 0)
                    hipLaunchKernelGGL(solo, dim3(grid), dim3(64), 0, 0, out, in);
                else
                    hipLaunchKernelGGL(duo, dim3(grid), dim3(128), 0, 0, out, in);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            hipEventElapsedTime(&ms[which], e0, e1);
        }
        printf("%d block(s) of 64 samples per SIMD, %d steps of 73 fp64 FMAs: one wavefront per block %.3f ms, soil + routing wavefronts "
               "%.3f ms (%+.1f %%)\n", per, STEPS, ms[0], ms[1], (ms[1] / ms[0] - 1) * 100);
    }
    return 0;
}
