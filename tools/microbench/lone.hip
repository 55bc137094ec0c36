// lone.hip -- what does ONE wavefront that has its SIMD to itself issue per cycle, and what decides it?
// (round 4: the time loops at 1e5 samples run at 1.5 wavefronts per SIMD; a 4-byte shift of a loop moved it by 4.7 %.)
// Every variant is one asm loop (the compiler has no say in it); its trip is timed in the wave itself with s_memtime
// (shader clock) and the waves of a grid are reported as a distribution, grouped by where they ran.
// Grids of 1, 2 and 3 waves per SIMD (1024 / 2048 / 3072 one-wave blocks on an idle chip land evenly:
// profiles/r01_microbench_wave_placement.txt).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define FMA(d) "v_fma_f64 " d ", " d ", v[20:21], v[22:23]\n\t"
#define S_ "s_add_u32 s20, s20, 1\n\t"
#define Q(X) X X X X
#define D1 FMA("v[0:1]") FMA("v[0:1]") FMA("v[0:1]") FMA("v[0:1]")             /* 4, one chain */
#define D4 FMA("v[0:1]") FMA("v[2:3]") FMA("v[4:5]") FMA("v[6:7]")             /* 4, four chains */
#define CLOB "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v20", "v21", "v22", "v23", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "scc", "vcc"

#define VARIANT(NAME, PAD, BODY)                                                                                    \
    __global__ __launch_bounds__(64) void NAME(unsigned long long *rec, const double *in, int iters)                \
    {                                                                                                               \
        prologue(in);                                                                                               \
        const unsigned long long c0 = __builtin_readcyclecounter();                                                 \
        asm volatile("s_mov_b32 s21, %0\n\t"                                                                        \
                     "s_mov_b32 s20, 0\n\t"                                                                         \
                     "s_mov_b64 s[22:23], 0\n\t"                                                                    \
                     "s_mov_b64 s[26:27], 15\n\t"                                                                    \
                     ".p2align 6\n\t" PAD "1:\n\t" BODY "s_sub_u32 s21, s21, 1\n\t"                                 \
                     "s_cmp_lg_u32 s21, 0\n\t"                                                                      \
                     "s_cbranch_scc1 1b\n\t" ::"s"(iters)                                                           \
                     : CLOB);                                                                                       \
        epilogue(rec, c0);                                                                                          \
    }

__device__ __forceinline__ void prologue(const double *in)
{
    asm volatile("v_mov_b32 v20, %0\n\tv_mov_b32 v21, %1\n\tv_mov_b32 v22, %2\n\tv_mov_b32 v23, %3\n\t"
                 "v_mov_b32 v0, 0\n\tv_mov_b32 v1, 0\n\tv_mov_b32 v2, 0\n\tv_mov_b32 v3, 0\n\t"
                 "v_mov_b32 v4, 0\n\tv_mov_b32 v5, 0\n\tv_mov_b32 v6, 0\n\tv_mov_b32 v7, 0\n\t" ::"v"(__double2loint(in[0])),
                 "v"(__double2hiint(in[0])), "v"(__double2loint(in[1])), "v"(__double2hiint(in[1]))
                 : CLOB);
}
__device__ __forceinline__ void epilogue(unsigned long long *rec, unsigned long long c0)
{
    const unsigned long long c1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) {
        rec[blockIdx.x * 3 + 0] = c1 - c0;
        rec[blockIdx.x * 3 + 1] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));     // HW_REG_HW_ID
        rec[blockIdx.x * 3 + 2] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));    // HW_REG_XCC_ID
    }
}

// one chain / four chains, the run on an 8-byte boundary / at 4 mod 8
VARIANT(dep64_a, "", Q(Q(D1)))
VARIANT(dep64_m, "s_nop 0\n\t", Q(Q(D1)))
VARIANT(ind64_a, "", Q(Q(D4)))
VARIANT(ind64_m, "s_nop 0\n\t", Q(Q(D4)))
// the trip's length: what the three scalar instructions and the taken branch at its end cost
VARIANT(ind16_a, "", Q(D4))
VARIANT(ind32_a, "", Q(D4) Q(D4))
VARIANT(ind128_a, "", Q(Q(D4)) Q(Q(D4)))
VARIANT(ind256_a, "", Q(Q(D4)) Q(Q(D4)) Q(Q(D4)) Q(Q(D4)))
// scalar instructions between the vector ones (pairs: the parity of what follows stays)
VARIANT(salu2_per4, "", Q(Q(D4 S_ S_)))
VARIANT(salu2_per8, "", Q(Q(D4 D4 S_ S_)) )
VARIANT(salu2_per16, "", Q(Q(D4) S_ S_))
VARIANT(nop2_per4, "", Q(Q(D4 "s_nop 0\n\ts_nop 0\n\t")))
// a conditional branch that is not taken (+ an s_nop for the parity) after every 8
VARIANT(brnt_per8, "", "s_cmp_lg_u64 s[22:23], 0\n\ts_nop 0\n\t" Q(Q(D4 D4 "s_cbranch_scc1 9f\n\ts_nop 0\n\t")) "9:\n\t")
// an unconditional branch over eight instructions after every 16 (the skipped ones are not counted)
VARIANT(brfw_per16, "", Q(Q(D4) "s_branch 8f\n\t s_nop 0\n\t" D4 D4 "8:\n\t"))
// ... the same, the target on a 64-byte line
VARIANT(brfw64_per16, "", Q(Q(D4) "s_branch 8f\n\t .p2align 6\n\t8:\n\t"))
// a vector compare + branch on vcc, not taken, after every 8 (v_cmp_gt_f64 VOP3 8 bytes; s_cbranch 4 + s_nop 4)
VARIANT(vcmp_per8, "", Q(Q(D4 D4 "v_cmp_gt_f64 vcc, v[20:21], v[20:21]\n\ts_cbranch_vccnz 9f\n\ts_nop 0\n\t")) "9:\n\t")
// a scalar compare + branch, not taken, after every 8
VARIANT(scmp_per8, "", Q(Q(D4 D4 "s_cmp_lg_u64 s[22:23], 0\n\ts_cbranch_scc1 9f\n\t")) "9:\n\t")


// ---- how a step loop can pick the code of its next step(s): a "dry step" here is 9 vector instructions ----------------
#define F9 D4 D4 FMA("v[0:1]")
#define SETUP_PC "s_getpc_b64 s[28:29]\n\t60:\n\t"
// as smart_fast_arms.h has it: two scalar compares and two conditional branches, none taken, ahead of every step
#define DISP2 "s_cmp_eq_u64 s[22:23], 0\n\ts_cbranch_scc0 99f\n\ts_cmp_eq_u64 s[26:27], 0\n\ts_cbranch_scc1 99f\n\t"
VARIANT(disp_2cmp_x4, "", DISP2 F9 DISP2 F9 DISP2 F9 DISP2 F9 "99:\n\t")
// one test of a precomputed class bit and one conditional branch, not taken
#define DISP1(j) "s_bitcmp1_b32 s26, " j "\n\ts_cbranch_scc0 99f\n\t"
VARIANT(disp_1bit_x4, "", DISP1("0") F9 DISP1("1") F9 DISP1("2") F9 DISP1("3") F9 "99:\n\t")
// a computed jump (base + offset, s_setpc_b64) to a block that lies elsewhere, ahead of every PAIR of steps
#define JUMP(l) "s_add_u32 s24, s28, " l "f-60b\n\ts_addc_u32 s25, s29, 0\n\ts_setpc_b64 s[24:25]\n\ts_nop 0\n\t.p2align 8\n\t" l ":\n\t"
VARIANT(jump_per2, SETUP_PC, JUMP("61") F9 F9 JUMP("62") F9 F9)
// ... ahead of every four steps
VARIANT(jump_per4, SETUP_PC, JUMP("61") F9 F9 F9 F9)
// ... ahead of every step
VARIANT(jump_per1, SETUP_PC, JUMP("61") F9 JUMP("62") F9 JUMP("63") F9 JUMP("64") F9)
// a direct unconditional branch to a block elsewhere ahead of every pair
#define BRANCH(l) "s_branch " l "f\n\ts_nop 0\n\t.p2align 8\n\t" l ":\n\t"
VARIANT(branch_per2, "", BRANCH("61") F9 F9 BRANCH("62") F9 F9)
// no dispatch at all: the floor
VARIANT(none_x4, "", F9 F9 F9 F9)

struct Case { const char *name; void (*fn)(unsigned long long *, const double *, int); int vec; const char *what; };

static void run(const Case &c, unsigned long long *rec, const double *in)
{
    const int iters = 160000 / c.vec;
    for (int wps : {1, 2, 3}) {
        const int grid = 1024 * wps;
        hipLaunchKernelGGL(c.fn, dim3(grid), dim3(64), 0, 0, rec, in, 10);
        (void)hipDeviceSynchronize();
        hipLaunchKernelGGL(c.fn, dim3(grid), dim3(64), 0, 0, rec, in, iters);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h((size_t)grid * 3);
        (void)hipMemcpy(h.data(), rec, h.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> t(grid);
        double by_simd[4] = {0, 0, 0, 0}, by_cu2[2] = {0, 0};
        int n_simd[4] = {0, 0, 0, 0}, n_cu2[2] = {0, 0};
        for (int b = 0; b < grid; ++b) {
            t[b] = (double)h[b * 3] / ((double)c.vec * iters);
            const unsigned hw = (unsigned)h[b * 3 + 1], simd = (hw >> 4) & 3, cu = (hw >> 8) & 15;
            by_simd[simd] += t[b]; n_simd[simd]++;
            by_cu2[cu & 1] += t[b]; n_cu2[cu & 1]++;
        }
        std::sort(t.begin(), t.end());
        double mean = 0;
        for (double x : t) mean += x;
        mean /= grid;
        printf("%-13s waves/SIMD=%d  cycles per vector instruction per wave: min %5.2f  p25 %5.2f  median %5.2f  p75 %5.2f  max %5.2f  "
               "mean %5.2f (per SIMD slot %5.2f) | by SIMD %5.2f %5.2f %5.2f %5.2f | even/odd CU %5.2f %5.2f\n",
               c.name, wps, t[0], t[grid / 4], t[grid / 2], t[3 * grid / 4], t[grid - 1], mean, mean / wps,
               by_simd[0] / n_simd[0], by_simd[1] / n_simd[1], by_simd[2] / n_simd[2], by_simd[3] / n_simd[3],
               by_cu2[0] / n_cu2[0], by_cu2[1] / n_cu2[1]);
    }
}

int main()
{
    unsigned long long *rec;
    double *in;
    (void)hipMalloc(&rec, 4096 * 3 * 8);
    (void)hipMalloc(&in, 64);
    double h[2] = {1.0000001, 1e-9};
    (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    const Case cases[] = {
        {"dep64_a", dep64_a, 64, ""}, {"dep64_m", dep64_m, 64, ""}, {"ind64_a", ind64_a, 64, ""}, {"ind64_m", ind64_m, 64, ""},
        {"ind16_a", ind16_a, 16, ""}, {"ind32_a", ind32_a, 32, ""}, {"ind128_a", ind128_a, 128, ""}, {"ind256_a", ind256_a, 256, ""},
        {"salu2_per4", salu2_per4, 64, ""}, {"salu2_per8", salu2_per8, 128, ""}, {"salu2_per16", salu2_per16, 64, ""},
        {"nop2_per4", nop2_per4, 64, ""}, {"brnt_per8", brnt_per8, 128, ""}, {"brfw_per16", brfw_per16, 64, ""},
        {"brfw64_per16", brfw64_per16, 64, ""}, {"vcmp_per8", vcmp_per8, 128, ""}, {"scmp_per8", scmp_per8, 128, ""},
        // per trip: four "dry steps" of 9 vector instructions; cycles per vector instruction x 9 = cycles per step
        {"none_x4", none_x4, 36, ""}, {"disp_2cmp_x4", disp_2cmp_x4, 36, ""}, {"disp_1bit_x4", disp_1bit_x4, 36, ""},
        {"jump_per1", jump_per1, 36, ""}, {"jump_per2", jump_per2, 36, ""}, {"jump_per4", jump_per4, 36, ""},
        {"branch_per2", branch_per2, 36, ""},
    };
    for (const Case &c : cases) run(c, rec, in);
    // where do the slow waves of a one-wave-per-SIMD grid sit?  One character per wave (cycles per vector instruction,
    // rounded), a group per CU in HW_ID order, a line per XCC and shader engine
    for (const Case &c : {cases[4], cases[21]})
        for (int grid : {1024, 512, 256}) {
            const int iters = 160000 / c.vec;
            hipLaunchKernelGGL(c.fn, dim3(grid), dim3(64), 0, 0, rec, in, iters);
            (void)hipDeviceSynchronize();
            std::vector<unsigned long long> h((size_t)grid * 3);
            (void)hipMemcpy(h.data(), rec, h.size() * 8, hipMemcpyDeviceToHost);
            printf("%s grid=%d\n", c.name, grid);
            std::vector<std::vector<int>> cell(8 * 8 * 16 * 4);
            for (int b = 0; b < grid; ++b) {
                const unsigned hw = (unsigned)h[b * 3 + 1], simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, se = (hw >> 13) & 7,
                               xcc = (unsigned)h[b * 3 + 2] & 7;
                cell[((xcc * 8 + se) * 16 + cu) * 4 + simd].push_back((int)((double)h[b * 3] / ((double)c.vec * iters) + 0.5));
            }
            for (int x = 0; x < 8; ++x)
                for (int se = 0; se < 8; ++se) {
                    bool any = false;
                    for (int i = 0; i < 64; ++i) any = any || !cell[(x * 8 + se) * 64 + i].empty();
                    if (!any) continue;
                    printf("  xcc %d se %d:", x, se);
                    for (int cu = 0; cu < 16; ++cu) {
                        printf(" ");
                        for (int sd = 0; sd < 4; ++sd) {
                            auto &v = cell[((x * 8 + se) * 16 + cu) * 4 + sd];
                            if (v.empty()) printf(".");
                            else for (int t : v) printf("%x", t > 15 ? 15 : t);
                        }
                    }
                    printf("\n");
                }
        }
    return 0;
}
