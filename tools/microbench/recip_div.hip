// recip_div.hip -- how often does the reciprocal-with-correction division of smart_literal_model.h (RECIP) miss the
// correctly rounded quotient?  a / b through y = RN(1 / b):
//   one correction   q0 = RN(a y); r0 = fma(-b, q0, a); q = fma(r0, y, q0)                          (round 3)
//   two corrections  ... q1 = fma(r0, y, q0); r1 = fma(-b, q1, a); q = fma(r1, y, q1)               (round 4)
// against the hardware's IEEE division, over random pairs: significands uniform in [1, 2) or drawn next to 1 and 2
// (where two roundings of relative size 2^-53 are up to two ulps of the quotient), exponents within +-200.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off recip_div.hip -o recip_div && ./recip_div [pairs per thread]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ inline unsigned long long next(unsigned long long &s)
{
    s ^= s << 13;
    s ^= s >> 7;
    s ^= s << 17;
    return s;
}

__device__ inline double draw(unsigned long long &s, int mode)
{
    const unsigned long long r = next(s);
    unsigned long long frac = r & 0x000fffffffffffffull;
    if (mode == 1)
        frac &= 0x3ff; // next to 1
    else if (mode == 2)
        frac |= 0x000ffffffffffc00ull; // next to 2
    const long e = 1023 + (long)((r >> 52) % 401) - 200;
    unsigned long long u = ((unsigned long long)e << 52) | frac;
    if ((u & 0x000fffffffffffffull) == 0x000fffffffffffffull)
        u ^= 1; // (divisors with a significand of all ones are excluded by the model as well)
    return __builtin_bit_cast(double, u);
}

__global__ void probe(long per_thread, unsigned long long seed, unsigned long long *counts, double *examples)
{
    unsigned long long s = seed + 0x9e3779b97f4a7c15ull * (blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x + 1);
    unsigned long long bad1 = 0, bad2 = 0, inexact = 0;
    for (long i = 0; i < per_thread; ++i) {
        const int mode_a = (int)(next(s) % 3), mode_b = (int)(next(s) % 3);
        const double a = draw(s, mode_a), b = draw(s, mode_b);
        const double y = 1.0 / b;
        const double want = a / b;
        const double q0 = a * y;
        const double r0 = __builtin_fma(-b, q0, a);
        const double q1 = __builtin_fma(r0, y, q0);
        const double r1 = __builtin_fma(-b, q1, a);
        const double q2 = __builtin_fma(r1, y, q1);
        if (q1 != want) {
            if (bad1 == 0 && atomicAdd(counts + 3, 1ull) < 8) {
                const unsigned long long k = atomicAdd(counts + 4, 1ull);
                if (k < 8) {
                    examples[k * 4] = a;
                    examples[k * 4 + 1] = b;
                    examples[k * 4 + 2] = q1;
                    examples[k * 4 + 3] = want;
                }
            }
            ++bad1;
        }
        bad2 += q2 != want;
        (void)inexact;
    }
    atomicAdd(counts + 0, bad1);
    atomicAdd(counts + 1, bad2);
    atomicAdd(counts + 2, (unsigned long long)per_thread);
}

int main(int argc, char **argv)
{
    const long per = argc > 1 ? atol(argv[1]) : 200000;
    unsigned long long *counts;
    double *ex;
    hipMalloc(&counts, 8 * sizeof(*counts));
    hipMalloc(&ex, 32 * sizeof(double));
    hipMemset(counts, 0, 8 * sizeof(*counts));
    hipLaunchKernelGGL(probe, dim3(4096), dim3(256), 0, 0, per, 0x1234567ull, counts, ex);
    unsigned long long h[8];
    double he[32];
    hipMemcpy(h, counts, sizeof(h), hipMemcpyDeviceToHost);
    hipMemcpy(he, ex, sizeof(he), hipMemcpyDeviceToHost);
    printf("%llu pairs: one correction step misses the IEEE quotient on %llu (%.3g of them), two steps on %llu\n", h[2], h[0],
           (double)h[0] / (double)h[2], h[1]);
    for (unsigned long long k = 0; k < (h[4] < 8 ? h[4] : 8); ++k)
        printf("  a = %a  b = %a  one step %a  division %a\n", he[k * 4], he[k * 4 + 1], he[k * 4 + 2], he[k * 4 + 3]);
    return 0;
}
