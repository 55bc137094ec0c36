// exec0.hip -- what does a vector instruction cost when EXEC is zero (a wave walking through the other side of a
// wave-uniform if/else without branching around it)?
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 20000
#define V8 "v_fma_f64 %0, %2, %3, %0\n v_fma_f64 %1, %2, %3, %1\n v_fma_f64 %0, %2, %3, %0\n v_fma_f64 %1, %2, %3, %1\n" \
           "v_fma_f64 %0, %2, %3, %0\n v_fma_f64 %1, %2, %3, %1\n v_fma_f64 %0, %2, %3, %0\n v_fma_f64 %1, %2, %3, %1\n"
template <int MODE>
__global__ __launch_bounds__(64) void k(double *out, const double *in)
{
    double a = in[0], b = in[1], x0 = in[2] + threadIdx.x, x1 = x0 + 1;
    for (int i = 0; i < ITER; ++i) {
        asm volatile(V8 V8 : "+v"(x0), "+v"(x1) : "v"(a), "v"(b));                      // 16 live instructions
        if (MODE == 1)                                                                   // + 16 with EXEC = 0
            asm volatile("s_mov_b64 s[20:21], exec\n s_mov_b64 exec, 0\n" V8 V8 "s_mov_b64 exec, s[20:21]\n"
                         : "+v"(x0), "+v"(x1) : "v"(a), "v"(b) : "s20", "s21");
        if (MODE == 2)                                                                   // + 16 more live
            asm volatile(V8 V8 : "+v"(x0), "+v"(x1) : "v"(a), "v"(b));
        if (MODE == 3)                                                                   // + a taken branch around 16
            asm volatile("s_branch 1f\n" V8 V8 "1:\n" : "+v"(x0), "+v"(x1) : "v"(a), "v"(b));
    }
    out[blockIdx.x * 64 + threadIdx.x] = x0 + x1;
}
template <int MODE>
void run(const char *what, double *out, double *in)
{
    for (int wps : {1, 2, 4}) {
        int grid = 1024 * wps;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(64), 0, 0, out, in);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(64), 0, 0, out, in);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s waves/SIMD=%d  %.3f ms  = %.1f ns per iteration per wave-slot\n", what, wps, ms, ms * 1e6 / ITER / wps);
        fflush(stdout);
    }
}
int main()
{
    double *out, *in;
    hipMalloc(&out, 4096 * 64 * 8); hipMalloc(&in, 64);
    double h[3] = {1.0000001, 1e-9, 0.5};
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    run<0>("16 v_fma_f64", out, in);
    run<1>("16 v_fma_f64 + 16 with EXEC=0", out, in);
    run<2>("32 v_fma_f64", out, in);
    run<3>("16 v_fma_f64 + taken branch over 16", out, in);
    return 0;
}
