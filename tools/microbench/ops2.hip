// ops2.hip -- issue cost of the remaining instruction kinds of the SMART step, 4 waves per SIMD (throughput regime)
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 20000
#define R8(X) X X X X X X X X
template <int KIND>
__global__ __launch_bounds__(64) void k(double *out, const double *in)
{
    double a = in[0], b = in[1], x0 = in[2] + threadIdx.x, x1 = x0 + 1;
    unsigned long long m = 0;
    for (int i = 0; i < ITER; ++i) {
        if (KIND == 0) asm volatile(R8("v_fma_f64 %0, %2, %3, %0\n v_fma_f64 %1, %2, %3, %1\n") : "+v"(x0), "+v"(x1) : "v"(a), "v"(b));
        if (KIND == 1) asm volatile(R8("v_cmp_le_f64 vcc, %2, %0\n v_cmp_le_f64 vcc, %3, %1\n") : "+v"(x0), "+v"(x1) : "v"(a), "v"(b) : "vcc");
        if (KIND == 2) asm volatile(R8("v_cmp_le_f64 %4, %2, %0\n v_cmp_le_f64 %4, %3, %1\n") : "+v"(x0), "+v"(x1) : "v"(a), "v"(b), "s"(m));
        if (KIND == 3) asm volatile(R8("v_max_f64 %0, -%0, 0\n v_max_f64 %1, -%1, 0\n") : "+v"(x0), "+v"(x1) : "v"(a), "v"(b));
        if (KIND == 4) asm volatile(R8("v_mul_f64 %0, %0, 0.5\n v_mul_f64 %1, %1, 0.5\n") : "+v"(x0), "+v"(x1) : "v"(a), "v"(b));
        if (KIND == 5) asm volatile(R8("v_ldexp_f64 %0, %0, -2\n v_ldexp_f64 %1, %1, 2\n") : "+v"(x0), "+v"(x1) : "v"(a), "v"(b));
        if (KIND == 6) asm volatile(R8("v_mov_b64 %0, %2\n v_mov_b64 %1, %3\n") : "+v"(x0), "+v"(x1) : "v"(a), "v"(b));
        if (KIND == 8) asm volatile(R8("v_fma_f64 %0, %2, s[20:21], %0\n v_fma_f64 %1, s[20:21], %3, %1\n") : "+v"(x0), "+v"(x1) : "v"(a), "v"(b) : "s20", "s21");
        if (KIND == 9) asm volatile(R8("v_add_f64 %0, %0, -%1\n v_min_f64 %1, %1, %0\n") : "+v"(x0), "+v"(x1) : "v"(a), "v"(b));
    }
    out[blockIdx.x * 64 + threadIdx.x] = x0 + x1;
}
template <int KIND>
void run(const char *what, double *out, double *in)
{
    const int wps = 4, grid = 1024 * wps;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND>), dim3(grid), dim3(64), 0, 0, out, in);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND>), dim3(grid), dim3(64), 0, 0, out, in);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s %.3f ms  %.2f ns per wave-instruction per SIMD\n", what, ms, ms * 1e6 / ITER / 16 / wps);
    fflush(stdout);
}
int main()
{
    double *out, *in;
    hipMalloc(&out, 4096 * 64 * 8); hipMalloc(&in, 64);
    double h[3] = {1.0000001, 1e-9, 0.5};
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    run<0>("v_fma_f64 (reference)", out, in);
    run<1>("v_cmp_le_f64 -> vcc", out, in);
    run<2>("v_cmp_le_f64 -> sgpr pair", out, in);
    run<3>("v_max_f64 with neg modifier and const", out, in);
    run<4>("v_mul_f64 by inline constant", out, in);
    run<5>("v_ldexp_f64", out, in);
    run<6>("v_mov_b64", out, in);
    run<8>("v_fma_f64 with an SGPR-pair operand", out, in);
    run<9>("v_add_f64 / v_min_f64 dependent pair", out, in);
    return 0;
}
