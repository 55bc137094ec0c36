// lanes.hip -- round 5: the literal step with one sample per LANE (LiteralModelT<true>, what smart_fast_illcond and the
// smartcpp.allsteps stand-in ran through round 4) against one sample per DPP ROW (LiteralLanesModel,
// smartpy_amd/csrc/smart_literal_lanes.h): cycles per step of a lone wavefront, and every bit of the final states and sums.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I smartpy_amd/csrc -I include -DSMART_LANES_COUNT -o tools/microbench/lanes tools/microbench/lanes.hip
//   tools/microbench/lanes [n_samples] [n_steps] [rk_hi]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>
#include "smart_literal_lanes.h"

using namespace smart;

#define CHECK(x)                                                                                                       \
    do {                                                                                                               \
        hipError_t e_ = (x);                                                                                           \
        if (e_ != hipSuccess) {                                                                                        \
            printf("%s: %s\n", #x, hipGetErrorString(e_));                                                             \
            exit(1);                                                                                                   \
        }                                                                                                              \
    } while (0)

constexpr int kOut = 30; // 19 variables, acc, num, den, slow steps, spare

template <class Model>
__device__ __forceinline__ void run(Model &m, const double2 *__restrict__ f, long T, double *o)
{
    double acc = 0.0, num = 0.0, den = 0.0;
    time_loop(m, f, T, [&](const double2 v, const double ex) { m.step(v.x, v.y, ex, acc, num, den); });
    double v[19];
    m.get_vars(v);
    for (int i = 0; i < 19; ++i)
        o[i] = v[i];
    o[19] = acc;
    o[20] = num;
    o[21] = den;
}

__device__ __forceinline__ void initial(const double *p, double area, double *st)
{
    for (int i = 0; i < 5; ++i)
        st[i] = 0.54 / 1000 * area / 8766 * p[6 + (i < 2 ? 0 : (i < 3 ? 1 : 2))] * 120.0;
    for (int i = 5; i < 11; ++i)
        st[i] = (p[5] / 12) / 1000 * area;
    st[11] = 0.54 / 1000 * area / 8766 * p[9] * 1200.0;
}

__global__ __launch_bounds__(64) void per_lane(long N, long T, double area, double dt, const double *params,
                                               const double2 *__restrict__ f, double *out, unsigned long long *cyc)
{
    long n = (long)blockIdx.x * 64 + threadIdx.x;
    const bool live = n < N;
    if (!live)
        n = N - 1;
    double p[10], st[12];
    for (int i = 0; i < 10; ++i)
        p[i] = params[n * 10 + i];
    LiteralModelT<true> m;
    m.setup(area, dt, p);
    initial(p, area, st);
    m.set_states(st);
    double o[kOut] = {};
    const unsigned long long c0 = __builtin_readcyclecounter();
    run(m, f, T, o);
    const unsigned long long c1 = __builtin_readcyclecounter();
    if (live)
        for (int i = 0; i < kOut; ++i)
            out[n * kOut + i] = o[i];
    if (threadIdx.x == 0)
        cyc[blockIdx.x] = c1 - c0;
}

__global__ __launch_bounds__(64) void per_row(long N, long T, double area, double dt, const double *params,
                                              const double2 *__restrict__ f, double *out, unsigned long long *cyc, int pad,
                                              unsigned *where)
{
    if ((int)blockIdx.x < pad) // (workgroups that return at once ahead of the working ones: the real grid's other classes)
        return;
    const long wg = (long)blockIdx.x - pad;
    long n = wg * 4 + (threadIdx.x >> 4);
    const bool live = n < N && (threadIdx.x & 15) == 0;
    if (n >= N)
        n = N - 1;
    double p[10], st[12];
    for (int i = 0; i < 10; ++i)
        p[i] = params[n * 10 + i];
    LiteralLanesModel m;
    m.setup(area, dt, p);
    initial(p, area, st);
    m.set_states(st);
    double o[kOut] = {};
    const unsigned long long c0 = __builtin_readcyclecounter();
    run(m, f, T, o);
    const unsigned long long c1 = __builtin_readcyclecounter();
#ifdef SMART_LANES_COUNT
    o[22] = (double)m.n_slow;
    for (int i = 0; i < 6; ++i)
        o[23 + i] = (double)m.n_why[i];
#endif
    if (live)
        for (int i = 0; i < kOut; ++i)
            out[n * kOut + i] = o[i];
    if (threadIdx.x == 0) {
        cyc[wg] = c1 - c0;
        where[wg] = (__builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11)) << 20) |      // HW_REG_XCC_ID
                    (__builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11)) & 0xfffff);      // HW_REG_HW_ID: SIMD 5:4, CU 11:8, SE 15:13
    }
}


// ---- what the instructions of the row form cost a lone wavefront: 64 of a kind in a row, 200 trips ----------------------
#define R16(X) X X X X X X X X X X X X X X X X
#define PROBE(NAME, BODY)                                                                                              \
    __global__ __launch_bounds__(64) void NAME(unsigned long long *rec, int iters)                                     \
    {                                                                                                                  \
        double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - 1e-12, c = 1e-30, d = 0.5;                                       \
        const unsigned long long c0 = __builtin_readcyclecounter();                                                    \
        for (int i = 0; i < iters; ++i)                                                                                \
            asm volatile(R16(BODY) R16(BODY) R16(BODY) R16(BODY) : "+v"(a), "+v"(d) : "v"(b), "v"(c));                 \
        const unsigned long long c1 = __builtin_readcyclecounter();                                                    \
        if (threadIdx.x == 0)                                                                                          \
            rec[0] = c1 - c0;                                                                                          \
        if (a == 123.0 && d == 7.0)                                                                                    \
            rec[1] = 1;                                                                                                \
    }
PROBE(p_fma_dep, "v_fma_f64 %0, %0, %2, %3\n\t")
PROBE(p_fmac_dep, "v_fmac_f64_e32 %0, %2, %3\n\t")
PROBE(p_fmac_dpp_dep, "v_fmac_f64_dpp %0, %2, %3 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t")
PROBE(p_fmac_dpp_bank, "v_fmac_f64_dpp %0, %2, %3 row_newbcast:3 row_mask:0xf bank_mask:0x4\n\t")
PROBE(p_fmac_dpp_2chains, "v_fmac_f64_dpp %0, %2, %3 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %2, %3 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t")
PROBE(p_mov_dpp, "v_mov_b64_dpp %0, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t")
PROBE(p_fill_pair, "v_fmac_f64_dpp %0, %2, %3 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_max_f64 %0, %0, 0\n\t")
PROBE(p_max_dep, "v_max_f64 %0, %0, 0\n\t")
PROBE(p_mul_dep, "v_mul_f64 %0, %0, %2\n\t")
PROBE(p_mul_2chains, "v_mul_f64 %0, %0, %2\n\tv_mul_f64 %1, %1, %2\n\t")
PROBE(p_rcp, "v_rcp_f64 %0, %0\n\t")
PROBE(p_div_scale, "v_div_scale_f64 %0, vcc, %0, %2, %0\n\t")
PROBE(p_div_fixup, "v_div_fixup_f64 %0, %0, %2, %3\n\t")
PROBE(p_salu, "s_add_u32 s20, s20, 1\n\t")

template <class K>
static void probe(const char *name, K kernel, int per_body)
{
    unsigned long long *d, h[2];
    CHECK(hipMalloc(&d, 16));
    const int iters = 200;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(kernel, dim3(1), dim3(64), 0, 0, d, iters);
        CHECK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
    }
    printf("%-20s %6.2f cycles per instruction\n", name, (double)h[0] / (iters * 64.0 * per_body));
    CHECK(hipFree(d));
}

static void probes()
{
    probe("v_fma_f64 dep", p_fma_dep, 1);
    probe("v_fmac_f64 dep", p_fmac_dep, 1);
    probe("v_fmac_f64_dpp dep", p_fmac_dpp_dep, 1);
    probe("v_fmac_f64_dpp bank", p_fmac_dpp_bank, 1);
    probe("v_fmac_f64_dpp x2", p_fmac_dpp_2chains, 2);
    probe("v_mov_b64_dpp", p_mov_dpp, 1);
    probe("fmac_dpp + v_max", p_fill_pair, 2);
    probe("v_max_f64 dep", p_max_dep, 1);
    probe("v_mul_f64 dep", p_mul_dep, 1);
    probe("v_mul_f64 x2", p_mul_2chains, 2);
    probe("v_rcp_f64 dep", p_rcp, 1);
    probe("v_div_scale_f64", p_div_scale, 1);
    probe("v_div_fixup_f64", p_div_fixup, 1);
    probe("s_add_u32", p_salu, 1);
}

static double uni(unsigned long long &s)
{
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    return (double)(s >> 11) / 9007199254740992.0;
}

int main(int argc, char **argv)
{
    if (argc > 1 && !strcmp(argv[1], "probe")) {
        probes();
        return 0;
    }
    const long N = argc > 1 ? atol(argv[1]) : 1160, T = argc > 2 ? atol(argv[2]) : 4018;
    const double rk_hi = argc > 3 ? atof(argv[3]) : 12.0;
    const int pad = argc > 5 ? atoi(argv[5]) : 0;
    const int mode = argc > 4 ? (!strcmp(argv[4], "wet") ? 1 : (!strcmp(argv[4], "dry") ? 2 : 0)) : 0; // every step wet / dry / as it comes
    const double area = 175.46e6, dt = 86400.0;
    const double lo[10] = {0.9, 0.0, 0.0, 0.0, 0.0, 15.0, 1.0, 48.0, 1200.0, 1.0};
    const double hi[10] = {1.1, 1.0, 0.3, 1.0, 0.013, 150.0, 240.0, 1440.0, 4800.0, rk_hi};
    unsigned long long seed = 12345;
    std::vector<double> params((size_t)N * 10), forcing((size_t)T * 2);
    for (long n = 0; n < N; ++n)
        for (int i = 0; i < 10; ++i)
            params[n * 10 + i] = lo[i] + (hi[i] - lo[i]) * uni(seed);
    for (long t = 0; t < T; ++t) { // daily rain on 80 % of the days (a crude gamma), PE on a yearly sine
        const bool wet = mode == 1 ? true : (mode == 2 ? false : uni(seed) < 0.8);
        const double g = -std::log(1.0 - uni(seed)) * 3.2 * uni(seed);
        forcing[2 * t] = wet ? g + (mode == 1 ? 6.0 : 0.0) : 0.0;
        forcing[2 * t + 1] = std::max(0.0, 1.47 * (1 + 0.85 * std::sin(2 * M_PI * ((double)(t % 365) - 110) / 365.25)));
    }
    double *d_par, *d_f, *d_out_a, *d_out_b;
    unsigned long long *d_cyc;
    const long blocks_a = (N + 63) / 64, blocks_b = (N + 3) / 4;
    CHECK(hipMalloc(&d_par, params.size() * 8));
    CHECK(hipMalloc(&d_f, forcing.size() * 8));
    CHECK(hipMalloc(&d_out_a, (size_t)N * kOut * 8));
    CHECK(hipMalloc(&d_out_b, (size_t)N * kOut * 8));
    CHECK(hipMalloc(&d_cyc, (size_t)blocks_b * 8));
    unsigned *d_where;
    CHECK(hipMalloc(&d_where, (size_t)blocks_b * 4));
    CHECK(hipMemcpy(d_par, params.data(), params.size() * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_f, forcing.data(), forcing.size() * 8, hipMemcpyHostToDevice));
    CHECK(hipMemset(d_out_a, 0xff, (size_t)N * kOut * 8));
    CHECK(hipMemset(d_out_b, 0xee, (size_t)N * kOut * 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    std::vector<unsigned long long> cyc(blocks_b);
    auto report = [&](const char *name, long blocks, float ms) {
        CHECK(hipMemcpy(cyc.data(), d_cyc, (size_t)blocks * 8, hipMemcpyDeviceToHost));
        std::sort(cyc.begin(), cyc.begin() + blocks);
        printf("%-9s %5ld waves  kernel %8.3f ms  cycles/step of a wave: min %7.1f median %7.1f max %7.1f\n", name,
               blocks, ms, (double)cyc[0] / T, (double)cyc[blocks / 2] / T, (double)cyc[blocks - 1] / T);
    };
    for (int rep = 0; rep < 3; ++rep) {
        float ms = 0;
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(per_lane, dim3((unsigned)blocks_a), dim3(64), 0, 0, N, T, area, dt, d_par, (const double2 *)d_f,
                           d_out_a, d_cyc);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        report("per_lane", blocks_a, ms);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(per_row, dim3((unsigned)(blocks_b + pad)), dim3(64), 0, 0, N, T, area, dt, d_par, (const double2 *)d_f,
                           d_out_b, d_cyc, pad, d_where);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        report("per_row", blocks_b, ms);
    }
    {   // how the working wavefronts of the row form were spread over the chip's SIMDs
        std::vector<unsigned> w(blocks_b);
        CHECK(hipMemcpy(w.data(), d_where, (size_t)blocks_b * 4, hipMemcpyDeviceToHost));
        std::vector<unsigned> key(blocks_b);
        for (long i = 0; i < blocks_b; ++i)
            key[i] = ((w[i] >> 20) << 16) | (((w[i] >> 13) & 7) << 8) | (((w[i] >> 8) & 15) << 4) | ((w[i] >> 4) & 3);
        std::sort(key.begin(), key.end());
        long simds = 0, worst = 0, run = 0;
        for (long i = 0; i < blocks_b; ++i) {
            run = (i && key[i] == key[i - 1]) ? run + 1 : 1;
            simds += run == 1;
            worst = std::max(worst, run);
        }
        printf("per_row: %ld wavefronts (behind %d that return at once) on %ld SIMDs, at most %ld on one\n", blocks_b, pad, simds, worst);
    }
    std::vector<double> a((size_t)N * kOut), b((size_t)N * kOut);
    CHECK(hipMemcpy(a.data(), d_out_a, a.size() * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(b.data(), d_out_b, b.size() * 8, hipMemcpyDeviceToHost));
    long differ = 0, rows = 0, slow = 0, why[6] = {};
    for (long n = 0; n < N; ++n) {
        bool bad = false;
        for (int i = 0; i < 22; ++i)
            if (std::memcmp(&a[n * kOut + i], &b[n * kOut + i], 8)) {
                if (differ < 12)
                    printf("  sample %ld field %d: per_lane %.17g per_row %.17g\n", n, i, a[n * kOut + i], b[n * kOut + i]);
                ++differ;
                bad = true;
            }
        rows += bad;
        slow += (long)b[n * kOut + 22];
        for (int i = 0; i < 6; ++i)
            why[i] += (long)b[n * kOut + 23 + i];
    }
    printf("chunks refused (x samples) for divisors %ld, forcing %ld, a layer out of range %ld, a layer above capacity %ld, the river %ld (%ld)\n",
           why[0], why[1], why[2], why[3], why[4], why[5]);
    printf("bits: %ld of %ld values differ (%ld of %ld samples); guarded steps taken by the row form: %ld of %ld\n", differ,
           N * 22, rows, N, slow, N * T);
    printf("sample 0: Q_out %.17g V_river %.17g acc %.17g\n", a[6], a[18], a[19]);
    return differ != 0;
}
