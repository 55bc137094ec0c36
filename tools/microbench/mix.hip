// mix.hip -- does scalar work of one wave overlap with the fp64 vector work of another wave on the same SIMD?
// Per iteration: 16 v_fma_f64 (2 independent chains) + K scalar instructions (s_add_u32) [+ B taken branches].
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 20000
#define V4 "v_fma_f64 %0, %3, %4, %0\n v_fma_f64 %1, %3, %4, %1\n v_fma_f64 %0, %3, %4, %0\n v_fma_f64 %1, %3, %4, %1\n"
#define S1 "s_add_u32 %2, %2, 1\n"
#define S4 S1 S1 S1 S1
template <int K, int BR>
__global__ __launch_bounds__(64) void k(double *out, const double *in)
{
    double a = in[0], b = in[1], x0 = in[2] + threadIdx.x, x1 = x0 + 1;
    unsigned s = 0;
    for (int i = 0; i < ITER; ++i) {
        if (K == 0) asm volatile(V4 V4 V4 V4 : "+v"(x0), "+v"(x1), "+s"(s) : "v"(a), "v"(b) : "scc");
        if (K == 4) asm volatile(V4 S1 V4 S1 V4 S1 V4 S1 : "+v"(x0), "+v"(x1), "+s"(s) : "v"(a), "v"(b) : "scc");
        if (K == 8) asm volatile(V4 S1 S1 V4 S1 S1 V4 S1 S1 V4 S1 S1 : "+v"(x0), "+v"(x1), "+s"(s) : "v"(a), "v"(b) : "scc");
        if (K == 16) asm volatile(V4 S4 V4 S4 V4 S4 V4 S4 : "+v"(x0), "+v"(x1), "+s"(s) : "v"(a), "v"(b) : "scc");
        if (BR) { // BR taken forward branches per iteration
            for (int j = 0; j < BR; ++j)
                asm volatile("s_branch 1f\n s_nop 0\n s_nop 0\n1:\n" ::: "memory", "scc");
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + s;
}
template <int K, int BR>
void run(double *out, double *in)
{
    for (int wps : {1, 2, 4}) {
        int grid = 1024 * wps;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL((k<K, BR>), dim3(grid), dim3(64), 0, 0, out, in);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<K, BR>), dim3(grid), dim3(64), 0, 0, out, in);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        fflush(stdout); printf("16 v_fma_f64 + %2d s_add + %d taken branches per iter | waves/SIMD=%d  %.3f ms  = %.1f ns per iteration per wave-slot (SIMD time / waves)\n",
               K, BR, wps, ms, ms * 1e6 / ITER / wps);
    }
}
int main()
{
    double *out, *in;
    hipMalloc(&out, 4096 * 64 * 8); hipMalloc(&in, 64);
    double h[3] = {1.0000001, 1e-9, 0.5};
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    run<0, 0>(out, in); run<4, 0>(out, in); run<8, 0>(out, in); run<16, 0>(out, in);
    run<0, 2>(out, in); run<0, 4>(out, in); run<8, 4>(out, in);
    return 0;
}
