// smart_device.h -- kernel argument block and the time-loop skeleton shared by the two math modes.
//
// Mapping (gfx950): one wavefront lane = one Monte-Carlo sample, one 64-thread workgroup = one wavefront,
// grid = (ceil(N / 64), n_catchments).  The whole warm-up + simulation time loop of
// structure.py:87-146 runs inside one launch; the 10 parameters and the 12 states of a sample stay in
// VGPRs, the forcing of a step is the same for every lane of a wavefront and is fetched with scalar
// loads (s_load, address uniform in blockIdx.y and the time index), report means / groundwater sums /
// objective-function moments are accumulated in registers, and the only per-step-scale HBM traffic is
// one coalesced 512-byte discharge store per wavefront per report step.
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

namespace smart {

constexpr int kMaxDiv = 12;
constexpr int kForcingInsane = 1 << 30; // some value of the catchment's forcing is negative, -0, infinite or NaN

struct KArgs {
    long N, T, W, gap, R, first_len; // first_len: steps in report interval 0 (raw mode with T % gap != 0)
    int report_type;                 // 1 summary, 2 raw
    int np_mean;                     // literal mode: reproduce numpy's pairwise mean through LDS (8 <= gap <= 128)
    double dt;
    const double *area;    // [C]
    const double *forcing; // [C][T][2]
    const double *params;  // [C?][N][10]
    long pstride_c;
    const double *extra;   // [C][7] | null
    const double *initial; // [C][N][12] | null
    const double *obs;     // [C][R] | null
    const double *gw_obs;  // [C] | null
    const double *ws;      // [C][8 + R]: n, mean(e), sum(e), sum((e-mean)^2), sum(e-mean), then e - mean per report step
    double *discharge;     // [C][R][ld] | null
    long ld;
    double *gw;         // [C][N]
    double *objfn;      // [C][N][8] | null
    double *final_vars; // [C][N][19] | null
    int exits = 1;            // interval engine with (1) or without (0) wave-uniform early exits, FastModel::kExits
    // which kernels this call launches (include/smart_amd.h, SMART_PLAN_*): a workgroup that meets a block nobody will
    // run (a stale plan) leaves SMART_STATUS_STALE_PLAN in the status word instead of a silent hole in the outputs
    int class_mask = 0xf;     // bit c: the kernel of arithmetic class c (wave_class) has been launched
    int pc_mask = 0x7;        // bit 0: the interval engine (forcing constant over each report interval), bit 1: the
                              // step loop, bit 2: the run engine (constant over runs of k steps, k a divisor of the gap)
    // workspace header (null without a workspace: no status word, no time slices, every wavefront scans the forcing)
    int *hdr = nullptr;       // [kHdrInts]: status word, ticket counters of the sliced kernels
    int *fflags = nullptr;    // [C]: forcing flags of catchment c (forcing_flags_of_step below), from smart_forcing_scan
    // a report every step (gap 1): the run as one stream of records (rain, PE, observation, deviation per step) and a
    // code word per pair of steps (SMART_A_EVERY_STREAM), from smart_forcing_scan; null: time_loop_arms_each
    const double *estream = nullptr;  // [C][every_pairs(T)][8]
    const unsigned *ecodes = nullptr; // [C][every_pairs(T)]
    int pair_stride = 0;              // bytes between the pair blocks of the kernel that will read the code words: the
                                      // models with the final state vector have larger blocks (SMART_PS_STRIDE)
    const uint2 *codes = nullptr; // [C][code_chunks(T)]: per chunk of four steps the two code words of the pair blocks
                                  // (SMART_A_PAIRS_STRETCH), from smart_forcing_scan; null: the threaded chunks
    // run lengths a catchment's forcing is tested for: the divisors of the report gap, largest first (div[0] = gap)
    int n_div = 0;
    int div[kMaxDiv] = {};
    // time-sliced launch (n_seg > 1): see "time-sliced launch" below
    int n_seg = 1;            // workgroups per block of 64 samples, each advancing one slice of the time axis
    long n_catch = 1;         // C
    long n_blocks = 0;        // ceil(N / 64), blocks per catchment
    long seg_blocks = 0;      // C * n_blocks
    double *seg_state = nullptr; // [seg_blocks][kSegFields][64] hand-over between a block's consecutive slices
    int *seg_flag = nullptr;     // [seg_blocks] slices completed (negative: the chain is poisoned, see wait_for_slice)
    long max_polls = 0;          // bound of a slice's wait for its predecessor, in polls of ~3 us
    int debug_drop = 0;          // test knob: slice 0 of block 0 never publishes (its successors must time out)
};

constexpr int kSegFields = 23; // 0-14 model states (13, 14: SPLIT only), 15 pending demand, 16 sum of outflows, 17-22 moments
constexpr int kHdrInts = 64;      // workspace header: [0] status word, [1 + k] ticket counter of sliced kernel k
constexpr int kHdrStatus = 0, kHdrTicket = 1;
constexpr int kStatusSliceTimeout = 1, kStatusStalePlan = 2, kStatusNonFiniteForcing = 4; // = SMART_STATUS_* of include/smart_amd.h

__device__ __forceinline__ void raise_status(const KArgs &a, int bit)
{
    if (a.hdr && threadIdx.x == 0)
        __hip_atomic_fetch_or(a.hdr + kHdrStatus, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}


#ifndef SMART_NT_STORE
#define SMART_NT_STORE 1
#endif
#ifndef SMART_IV_DEFER
#define SMART_IV_DEFER 0 // interval / run engine: 1 = a dry interval defers its evaporation demand to the next wet one (measured:
                         // no gain at any load, and its extra live value costs the kernel with exits its third wave per SIMD)
#endif
#ifndef SMART_CHUNK_THREADED
#define SMART_CHUNK_THREADED 1 // the four steps of a chunk as one threaded asm (0: four single-step asms; A/B builds)
#endif
// code placement of a hot loop: onto a 64-byte line, `phase` dwords (s_nop) behind it
#define SMART_NOPS_0 ""
#define SMART_NOPS_1 "s_nop 0\n\t"
#define SMART_NOPS_2 SMART_NOPS_1 SMART_NOPS_1
#define SMART_NOPS_4 SMART_NOPS_2 SMART_NOPS_2
#define SMART_NOPS_8 SMART_NOPS_4 SMART_NOPS_4
#define SMART_NOPS_3 SMART_NOPS_2 SMART_NOPS_1
#define SMART_NOPS_5 SMART_NOPS_4 SMART_NOPS_1
#define SMART_NOPS_6 SMART_NOPS_4 SMART_NOPS_2
#define SMART_NOPS_7 SMART_NOPS_4 SMART_NOPS_3
#define SMART_NOPS_9 SMART_NOPS_8 SMART_NOPS_1
#define SMART_NOPS_10 SMART_NOPS_8 SMART_NOPS_2
#define SMART_NOPS_11 SMART_NOPS_8 SMART_NOPS_3
#define SMART_NOPS_12 SMART_NOPS_8 SMART_NOPS_4
#define SMART_NOPS_13 SMART_NOPS_8 SMART_NOPS_5
#define SMART_NOPS_14 SMART_NOPS_8 SMART_NOPS_6
#define SMART_NOPS_15 SMART_NOPS_8 SMART_NOPS_7
#define SMART_NOPS_(n) SMART_NOPS_##n
#define SMART_NOPS(n) SMART_NOPS_(n)
#define SMART_PLACE_LOOP(phase) asm volatile(".p2align 6\n\t" SMART_NOPS(phase))
#ifndef SMART_STEPS_PHASE
#define SMART_STEPS_PHASE 1
#endif
#ifndef SMART_PAIR_BLOCKS
#define SMART_PAIR_BLOCKS 1 // the streaming step loop as pair blocks behind computed jumps (0: the threaded chunks; A/B builds)
#endif
#ifndef SMART_PS_STRIDE
#define SMART_PS_STRIDE 2368 // ... of the SPLIT models (two more reservoirs in every arm)
#endif
#ifndef SMART_E_STRIDE
#define SMART_E_STRIDE 2240 // bytes from one block of the every-step stream to the next (smart_fast_arms.h)
#endif
#ifndef SMART_EVERY_PHASE
#define SMART_EVERY_PHASE 2 // dwords between a 64-byte line and the loop of time_loop_arms_each
#endif
#ifndef SMART_P_STRIDE
#define SMART_P_STRIDE 2112 // bytes from one pair block to the next (smart_fast_arms.h; a multiple of 64, not a power of two)
#endif
#ifndef SMART_STEP_ARMS
#define SMART_STEP_ARMS 1 // the step loop of sub-daily forcing as three asm arms (0: the compiled step_lazy of round 2)
#endif

constexpr int kWave = 64;
constexpr int kIllCondWaves = 16; // workgroups per block of 64 samples of the kernels that take one sample per DPP row

// NaN test on the bit pattern (missing observation / absent constraint): survives -fno-honor-nans, and for the
// wave-uniform values it is applied to it is scalar integer work.
__device__ __forceinline__ bool is_nan_bits(double x)
{
    return (__builtin_bit_cast(unsigned long long, x) & 0x7fffffffffffffffull) > 0x7ff0000000000000ull;
}
constexpr int kWsHead = 8;

// What smart_obs_prepare writes for the deviation e - mean of a MISSING observation: a NaN whose payload no arithmetic
// produces (a computed NaN is the canonical 0x7ff8000000000000, or carries the payload of an input NaN -- and an
// observation that is a NaN is missing whatever its payload).  A report every step tells a missing observation from the
// upper half of the deviation it has in a scalar register anyway: one s_cmp_eq_u32 (the exact NaN test of the
// observation itself in 32-bit pieces was eleven scalar instructions a step in hipcc's hands).
constexpr unsigned long long kMissingObs = 0x7ff8dead00000000ull;
__device__ __forceinline__ bool is_missing_mark(double w)
{
    return (unsigned)(__builtin_bit_cast(unsigned long long, w) >> 32) == (unsigned)(kMissingObs >> 32);
}

// a quiet NaN that does not trip -fno-honor-nans diagnostics (the fast kernels never do arithmetic on one)
__host__ __device__ __forceinline__ double quiet_nan() { return __builtin_bit_cast(double, 0x7ff8000000000000ull); }

// objective functions from the one-pass moments (montecarlo.py:193-209; formulas of spotpy's nashsutcliffe,
// kge(return_all=True), pbias, rmse).  Moments are taken about the observation mean, known before the run:
//   A = sum(s - e)   B = sum((s - e)^2)   C1 = sum(s - c)   C2 = sum((s - c)^2)   C3 = sum((e - ebar)(s - c))
// with ebar the observation mean and c ANY constant per sample (variance and covariance do not depend on it); the
// kernels take c = the sample's own first reported discharge: a simulated series that barely moves (std 1e-4 of its
// mean) then keeps its variance through the one-pass form, where c = ebar lost half the digits (KGEa off by 2e-7)
__device__ inline void finish_objectives(const double *st, double A, double B, double C1, double C2, double C3,
                                         double gw_sim, double gw_obs, double *o)
{
    const double n = st[0], se = st[2], see = st[3], sd = st[4];
    const double inv_n = 1.0 / n;
    const double m1 = C1 * inv_n;
    const double var_s = C2 * inv_n - m1 * m1;
    const double var_e = see * inv_n;
    const double cov = (C3 - sd * m1) * inv_n;
    double cc = cov / sqrt(var_s * var_e);
    // np.corrcoef clips to [-1, 1]; a NaN (constant observations or a single one: 0 / 0) stays a NaN there, which
    // fmin / fmax would turn into -1 -- tested on the bit pattern, the translation units assume NaN-free arithmetic
    if (!is_nan_bits(cc))
        cc = fmin(fmax(cc, -1.0), 1.0);
    const double alpha = sqrt(var_s / var_e);
    const double beta = 1.0 + A / se;
    o[0] = 1.0 - B / see;
    o[1] = 1.0 - sqrt((cc - 1.0) * (cc - 1.0) + (alpha - 1.0) * (alpha - 1.0) + (beta - 1.0) * (beta - 1.0));
    o[2] = cc;
    o[3] = alpha;
    o[4] = beta;
    o[5] = 100.0 * (A / se);
    o[6] = sqrt(B * inv_n);
    if (!is_nan_bits(gw_obs)) // objfunctions.py:20-24
        o[7] = (gw_obs - 0.1 <= gw_sim && gw_sim <= gw_obs + 0.1) ? 1.0 : 0.0;
    else
        o[7] = quiet_nan();
}

// numpy's pairwise sum of n (< 8, or 8..128) values held in LDS column `lane` (stride 64 doubles):
// the order np.mean(np.reshape(Q_out, (-1, gap)), axis=-1) adds them in (structure.py:190).
__device__ inline double np_pairwise_lds(const double *col, long n)
{
    if (n < 8) {
        double r = 0.0;
        for (long i = 0; i < n; ++i)
            r += col[i * kWave];
        return r;
    }
    double r0 = col[0], r1 = col[kWave], r2 = col[2 * kWave], r3 = col[3 * kWave];
    double r4 = col[4 * kWave], r5 = col[5 * kWave], r6 = col[6 * kWave], r7 = col[7 * kWave];
    long i = 8;
    for (; i < n - (n % 8); i += 8) {
        const double *b = col + i * kWave;
        r0 += b[0];
        r1 += b[kWave];
        r2 += b[2 * kWave];
        r3 += b[3 * kWave];
        r4 += b[4 * kWave];
        r5 += b[5 * kWave];
        r6 += b[6 * kWave];
        r7 += b[7 * kWave];
    }
    double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; ++i)
        res += col[i * kWave];
    return res;
}

// Step kinds: a generic step decides wet / dry per lane; StepAllWet is used by the interval loop when every lane of
// the wavefront is on the wet side for a whole report interval (no compare, no branch).
struct StepGeneric {};
struct StepAllWet {};

__device__ __forceinline__ bool same_bits(double x, double y)
{
    return __builtin_bit_cast(long long, x) == __builtin_bit_cast(long long, y);
}

// Walk n time steps of one catchment's forcing.  The forcing of a step is the same for all 64 lanes, so it is
// fetched with scalar loads into SGPRs: kChunk steps (one 64-byte line) per s_load_dwordx16, and the next chunk is
// requested before the current one is consumed, so the load latency hides behind kChunk model steps.
#ifndef SMART_CHUNK
#define SMART_CHUNK 4
#endif
constexpr int kChunk = SMART_CHUNK;

// does the model want to look at a chunk before its steps (LiteralModelT<true>::begin_chunk)?
template <class Model, class = void>
struct chunk_hook : std::false_type {};
template <class Model>
struct chunk_hook<Model, std::enable_if_t<Model::kChunkHook>> : std::true_type {};

// ... and does its step come in two forms chosen per chunk (LiteralLanesModel::quick)?
template <class Model, class = void>
struct chunk_modes : std::false_type {};
template <class Model>
struct chunk_modes<Model, std::enable_if_t<Model::kChunkModes>> : std::true_type {};

// The rain excess of the kChunk steps of a chunk is evaluated together, ahead of the steps (independent FMAs that
// fill issue slots while the first step's dependent chain starts).
template <class Model, class Body, class ChunkEnd>
__device__ __forceinline__ void time_loop_chunked(const Model &m, const double2 *__restrict__ f, long n, Body &&body,
                                                  ChunkEnd &&chunk_end)
{
    const long n_chunks = n / kChunk;
    double2 cur[kChunk], nxt[kChunk];
    if (n_chunks > 0) {
#pragma unroll
        for (int j = 0; j < kChunk; ++j)
            cur[j] = f[j];
    }
    for (long ch = 0; ch < n_chunks; ++ch) {
        const long pre = (ch + 1 < n_chunks ? ch + 1 : ch) * kChunk; // last chunk: harmless re-load of itself
#pragma unroll
        for (int j = 0; j < kChunk; ++j)
            nxt[j] = f[pre + j];
        if constexpr (chunk_hook<Model>::value)
            const_cast<Model &>(m).begin_chunk(cur, kChunk);
        double ex[kChunk];
#pragma unroll
        for (int j = 0; j < kChunk; ++j)
            ex[j] = m.excess(cur[j].x, cur[j].y);
        if constexpr (chunk_modes<Model>::value) {
            // the model's step comes in two forms and begin_chunk() has just said which one this chunk takes (wave-uniform,
            // unchanged by the steps): decided here, once, the four steps carry no test of their own -- a lone wavefront
            // pays 16 cycles for a branch it does not take (profiles/r04_microbench_lone.txt)
            if (__builtin_expect(m.quick, 1)) {
#pragma unroll
                for (int j = 0; j < kChunk; ++j)
                    body(cur[j], ex[j]);
            } else {
#pragma unroll
                for (int j = 0; j < kChunk; ++j)
                    body(cur[j], ex[j]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < kChunk; ++j)
                body(cur[j], ex[j]);
        }
        chunk_end();
#pragma unroll
        for (int j = 0; j < kChunk; ++j)
            cur[j] = nxt[j];
    }
    for (long t = n_chunks * kChunk; t < n; ++t) {
        const double2 v = f[t];
        if constexpr (chunk_hook<Model>::value)
            const_cast<Model &>(m).begin_chunk(&v, 1);
        body(v, m.excess(v.x, v.y));
    }
}

template <class Model, class Body>
__device__ __forceinline__ void time_loop(const Model &m, const double2 *__restrict__ f, long n, Body &&body)
{
    time_loop_chunked(m, f, n, body, [] {});
}

// The same walk for the step loop with deferred evaporation (Model::step_lazy), which wants to know, as a lane mask
// in scalar registers, which lanes are on the wet side of each step (ex >= 0): the kChunk compares of a chunk are
// issued together with its excesses, ahead of the first step, so that the branches of a step depend on scalar
// registers written long before and not on a vector compare the wave would have to wait for (-2.7 % on the bench's
// sub-daily forcing, same-box A/B).
// (Tried on top, measured, dropped: straight-line arms for chunks whose steps are dry -- or calm -- for every lane,
// to spare them the taken branch per step.  The three-way join made hipcc carry the layers in two register sets and
// copy between them: 135 v_mov_b64 in the loop instead of 61, 20.1 ms instead of 18.1.)
template <class Model, class Calm, class ChunkEnd>
__device__ __forceinline__ void time_loop_lazy(Model &m, const double2 *__restrict__ f, long n, Calm &&calm,
                                               double &acc, double &num, double &den, ChunkEnd &&chunk_end)
{
    const long n_chunks = n / kChunk;
    double2 cur[kChunk], nxt[kChunk];
    if (n_chunks > 0) {
#pragma unroll
        for (int j = 0; j < kChunk; ++j)
            cur[j] = f[j];
    }
    for (long ch = 0; ch < n_chunks; ++ch) {
        const long pre = (ch + 1 < n_chunks ? ch + 1 : ch) * kChunk; // last chunk: harmless re-load of itself
#pragma unroll
        for (int j = 0; j < kChunk; ++j)
            nxt[j] = f[pre + j];
        double ex[kChunk];
        unsigned long long wet[kChunk];
#pragma unroll
        for (int j = 0; j < kChunk; ++j)
            ex[j] = m.excess(cur[j].x, cur[j].y);
#pragma unroll
        for (int j = 0; j < kChunk; ++j)
            wet[j] = __builtin_amdgcn_ballot_w64(ex[j] >= 0.0);
#pragma unroll
        for (int j = 0; j < kChunk; ++j)
            m.step_lazy(ex[j], wet[j], calm(cur[j]), acc, num, den);
        chunk_end();
#pragma unroll
        for (int j = 0; j < kChunk; ++j)
            cur[j] = nxt[j];
    }
    for (long t = n_chunks * kChunk; t < n; ++t) {
        const double2 v = f[t];
        const double ex = m.excess(v.x, v.y);
        m.step_lazy(ex, __builtin_amdgcn_ballot_w64(ex >= 0.0), calm(v), acc, num, den);
    }
}

// The walk of the instruction-level step loop (Model::step_arms: one of three asm arms per step, picked on the scalar
// unit from the step's own forcing).  Same double-buffered scalar loads; the loop counter is 32 bits wide (there is no
// 64-bit scalar less-than: hipcc went through a vector compare for it).
template <bool QUICK, bool LAST = false, class Model, class ChunkEnd>
__device__ __forceinline__ void time_loop_arms(Model &m, const double2 *__restrict__ f, long n, double &acc,
                                               ChunkEnd &&chunk_end)
{
    const int n_chunks = (int)(n / kChunk);
    double2 cur[kChunk], nxt[kChunk];
    if (n_chunks > 0) {
#pragma unroll
        for (int j = 0; j < kChunk; ++j)
            cur[j] = f[j];
        // wait for the first chunk HERE: left pending into the loop, its s_waitcnt lands at the top of the body, behind
        // the request for the next chunk -- and then waits for that one too, every iteration
#pragma unroll
        for (int j = 0; j < kChunk; ++j)
            asm volatile("" ::"s"(cur[j].x), "s"(cur[j].y));
    }
    for (int ch = 0; ch < n_chunks; ++ch) {
        const int pre = (ch + 1 < n_chunks ? ch + 1 : ch) * kChunk; // last chunk: harmless re-load of itself
#pragma unroll
        for (int j = 0; j < kChunk; ++j)
            nxt[j] = f[pre + j];
#if SMART_CHUNK_THREADED
        static_assert(kChunk == 4, "SMART_A_CHUNK threads four steps");
        m.template chunk_arms<QUICK, LAST>(cur, acc);
#else
#pragma unroll
        for (int j = 0; j < kChunk; ++j)
            m.template step_arms<QUICK, LAST>(cur[j], acc);
#endif
        chunk_end();
#pragma unroll
        for (int j = 0; j < kChunk; ++j)
            cur[j] = nxt[j];
    }
    for (long t = (long)n_chunks * kChunk; t < n; ++t)
        m.template step_arms<QUICK, LAST>(f[t], acc);
}

// Data that nothing writes during the launch (observations, their deviations), read through the CONSTANT address space:
// a uniform load from there is a scalar load whatever stores the loop around it makes -- __restrict__ on a pointer that
// has passed through a struct or a lambda does not get hipcc that far.
typedef const double __attribute__((address_space(4))) *const_f64;
__device__ __forceinline__ const_f64 as_constant(const double *p) { return (const_f64)(unsigned long long)p; }

// The same walk with a report after EVERY step (gap 1): single-step arms with the routing of SMART_A_ROUTE_LAST, so that
// `acc` holds the outflow of the step just taken when step_end(t, e, w) looks at it.  OBS: the observation e of each
// step's report and its deviation w from the mean travel with the forcing -- scalar loads, a chunk of four steps ahead
// (left to the reporter, each step waits for two VECTOR loads of its own, issued behind the step: hipcc cannot prove
// that the discharge stores leave the observations alone once the pointers live in a struct; that alone was a third of
// the first version's 31.5 ms at 1e5 samples).  obs / dev point at the first step's entries.
template <bool QUICK, bool OBS, class Model, class StepEnd>
__device__ __forceinline__ void time_loop_arms_each(Model &m, const double2 *__restrict__ f, const double *obs_,
                                                    const double *dev_, long n, double &acc, StepEnd &&step_end)
{
    const const_f64 obs = as_constant(obs_), dev = as_constant(dev_);
    const int n_chunks = (int)(n / kChunk);
    double2 cur[kChunk], nxt[kChunk];
    double e_cur[kChunk] = {}, e_nxt[kChunk] = {}, w_cur[kChunk] = {}, w_nxt[kChunk] = {};
    if (n_chunks > 0) {
#pragma unroll
        for (int j = 0; j < kChunk; ++j) {
            cur[j] = f[j];
            if constexpr (OBS) {
                e_cur[j] = obs[j];
                w_cur[j] = dev[j];
            }
        }
#pragma unroll
        for (int j = 0; j < kChunk; ++j)
            asm volatile("" ::"s"(cur[j].x), "s"(cur[j].y));
    }
    // (where the loop lies counts: SMART_PLACE_LOOP; the phase by measurement, tools/gpu_round.sh phases)
    if (n_chunks > 0)
        SMART_PLACE_LOOP(SMART_EVERY_PHASE);
    for (int ch = 0; ch < n_chunks; ++ch) {
        const int pre = (ch + 1 < n_chunks ? ch + 1 : ch) * kChunk; // last chunk: harmless re-load of itself
#pragma unroll
        for (int j = 0; j < kChunk; ++j) {
            nxt[j] = f[pre + j];
            if constexpr (OBS) {
                e_nxt[j] = obs[pre + j];
                w_nxt[j] = dev[pre + j];
            }
        }
#pragma unroll
        for (int j = 0; j < kChunk; ++j) {
            m.template step_arms<QUICK, true>(cur[j], acc);
            step_end((long)ch * kChunk + j, e_cur[j], w_cur[j]);
        }
#pragma unroll
        for (int j = 0; j < kChunk; ++j) {
            cur[j] = nxt[j];
            e_cur[j] = e_nxt[j];
            w_cur[j] = w_nxt[j];
        }
    }
    for (long t = (long)n_chunks * kChunk; t < n; ++t) {
        m.template step_arms<QUICK, true>(f[t], acc);
        if constexpr (OBS)
            step_end(t, obs[t], dev[t]);
        else
            step_end(t, 0.0, 0.0);
    }
}

// A stretch of `n_iv` whole report intervals of `gap` steps, walked by the arm loop; interval_end() after each.
// A lone wavefront issues ONE instruction of any kind per turn of its SIMD (every fourth cycle): a scalar instruction
// or a branch costs it what a vector instruction costs, so the glue around the chunks counts -- the round-2 shape (one
// flat loop over chunks, the report test on chunk boundaries, an index clamped for the prefetch) spent 6.5 scalar
// instructions per STEP on it.  Here the chunk loop's own counter doubles as the interval counter, the prefetch
// address is a pointer that is only ever incremented, and nothing is clamped: the stretch streams through all its
// intervals but -- when it ends where the catchment's forcing ends -- the last one, which the clamped loop above
// takes (a prefetch must never read past the array).
template <bool QUICK, bool LAST = false, class Model, class IntervalEnd>
__device__ __forceinline__ void arm_intervals(Model &m, const double2 *__restrict__ f, long n_iv, long gap,
                                              bool ends_at_array_end, double &acc, IntervalEnd &&interval_end)
{
    const int cpi = (int)(gap / kChunk);
    long n_stream = 0;
#if SMART_CHUNK_THREADED
    if (gap % (2 * kChunk) == 0 && n_iv > 0) // (an even number of chunks per interval: the two buffers swap roles)
        n_stream = ends_at_array_end ? n_iv - 1 : n_iv;
#endif
    if (n_stream > 0) {
        double2 cur[kChunk], nxt[kChunk];
#pragma unroll
        for (int j = 0; j < kChunk; ++j)
            cur[j] = f[j];
#pragma unroll
        for (int j = 0; j < kChunk; ++j)
            asm volatile("" ::"s"(cur[j].x), "s"(cur[j].y)); // wait here, not behind the first prefetch (see above)
        const double2 *__restrict__ p = f;
        // s_waitcnt for a chunk HERE: scalar loads return out of order, so the only wait there is waits for everything
        // in flight -- it has to come before the next request goes out, not behind it
        auto arrived = [](const double2(&c)[kChunk]) {
#pragma unroll
            for (int j = 0; j < kChunk; ++j)
                asm volatile("" ::"s"(c[j].x), "s"(c[j].y));
        };
        // Two chunks per turn, the two buffers swapping roles: no copy from one to the other (the clamped loop above
        // pays 8 scalar moves a chunk for that).  Every half turn: wait for the chunk at hand, request the next one
        // into the other buffer, run the chunk.  11 scalar instructions per 8 steps.
        // WHERE the loop lies counts (round 4): the two chunks are 6 KB of code with two dozen branch targets, and their
        // phase against the 64-byte instruction-cache lines moved the launch by 5 % when an edit in a cold function
        // shifted it (13.85 -> 14.5 ms at 1e5 samples; same instructions, same registers).  The loop is therefore put on
        // a line boundary of its own, SMART_STEPS_PHASE dwords behind it: the phase measured best (tools/gpu_round.sh
        // phases), and no longer a matter of what is compiled in front of it.  Executed once per stretch.
        SMART_PLACE_LOOP(SMART_STEPS_PHASE);
        for (long iv = 0; iv < n_stream; ++iv) {
#pragma nounroll
            for (int c = 0; c < cpi; c += 2) {
                arrived(cur);
                p += kChunk;
#pragma unroll
                for (int j = 0; j < kChunk; ++j)
                    nxt[j] = p[j];
                m.template chunk_arms<QUICK, LAST>(cur, acc);
                arrived(nxt);
                p += kChunk;
#pragma unroll
                for (int j = 0; j < kChunk; ++j)
                    cur[j] = p[j];
                m.template chunk_arms<QUICK, LAST>(nxt, acc);
            }
            interval_end();
        }
    }
    for (long iv = n_stream; iv < n_iv; ++iv) {
        time_loop_arms<QUICK, LAST>(m, f + iv * gap, gap, acc, [] {});
        interval_end();
    }
}

// ---- pieces shared by the two launch bodies below ------------------------------------------------------------
struct LaneCtx {
    int lane;
    long c, n; // catchment, sample (clamped to N - 1 for the lanes of the last wavefront beyond the batch)
    bool live;
};

__device__ __forceinline__ LaneCtx lane_ctx(const KArgs &a, long block, long catchment)
{
    LaneCtx x;
    x.lane = threadIdx.x;
    x.c = catchment;
    x.n = block * kWave + x.lane;
    x.live = x.n < a.N;
    if (!x.live)
        x.n = a.N - 1;
    return x;
}

__device__ __forceinline__ LaneCtx lane_ctx(const KArgs &a) { return lane_ctx(a, (long)blockIdx.x, (long)blockIdx.y); }

// ... for the models that spread ONE sample over the sixteen lanes of a DPP row (smart_literal_lanes.h): a block of 64
// samples is sixteen wavefronts, wavefront `sub` of the block takes its samples 4 sub .. 4 sub + 3, one per row, and
// the first lane of a row does the row's stores
__device__ __forceinline__ LaneCtx lane_ctx_rows(const KArgs &a, long block, long catchment, int sub)
{
    LaneCtx x;
    x.lane = threadIdx.x;
    x.c = catchment;
    x.n = block * kWave + sub * 4 + (x.lane >> 4);
    x.live = x.n < a.N && (x.lane & 15) == 0;
    if (x.n >= a.N)
        x.n = a.N - 1;
    return x;
}

template <class Model, class = void>
struct lanes_per_sample : std::integral_constant<int, 1> {};
template <class Model>
struct lanes_per_sample<Model, std::enable_if_t<(Model::kLanesPerSample > 1)>>
    : std::integral_constant<int, Model::kLanesPerSample> {};

// parameters, derived constants and the initial states of structure.py:97-140 (educated guess / given states)
template <class Model>
__device__ __forceinline__ void init_model(const KArgs &a, const LaneCtx &x, Model &m)
{
    double p[10];
    {
        const double *pp = a.params + x.c * a.pstride_c + x.n * 10;
#pragma unroll
        for (int i = 0; i < 10; ++i)
            p[i] = pp[i];
    }
    const double area = a.area[x.c];
    m.setup(area, a.dt, p);

    double st[12];
    if (a.initial) {
        const double *ip = a.initial + (x.c * a.N + x.n) * 12;
#pragma unroll
        for (int i = 0; i < 12; ++i)
            st[i] = ip[i];
    } else {
#pragma unroll
        for (int i = 0; i < 12; ++i)
            st[i] = 0.0;
        if (a.extra) { // structure.py:100-112 / :125-137, same operation order
            const double *e = a.extra + x.c * 7;
            const double ro = e[0] * e[1];
            st[0] = ro * e[2] / 1000 * area / 8766 * p[6];
            st[1] = ro * e[3] / 1000 * area / 8766 * p[6];
            st[2] = ro * e[4] / 1000 * area / 8766 * p[7];
            st[3] = ro * e[5] / 1000 * area / 8766 * p[8];
            st[4] = ro * e[6] / 1000 * area / 8766 * p[8];
            st[11] = ro / 1000 * area / 8766 * p[9];
        }
        const double half = (p[5] / 12) / 1000 * area; // structure.py:115-116 / :139-140
#pragma unroll
        for (int i = 5; i < 11; ++i)
            st[i] = half;
    }
    m.set_states(st);
}

// what happens at the end of a report interval: the discharge store and the objective-function moments
struct Reporter {
    const double *__restrict__ obs; // this catchment's observations | null
    const double *__restrict__ ws;  // this catchment's workspace (statistics + e - mean) | null
    double shift; // c above: this sample's discharge of report step 0
    bool want_obj;
    double A = 0.0, B = 0.0, C1 = 0.0, C2 = 0.0, C3 = 0.0;

    __device__ __forceinline__ void init(const KArgs &a, const LaneCtx &x, const double *__restrict__ obs_all,
                                         const double *__restrict__ ws_all)
    {
        want_obj = a.objfn != nullptr;
        ws = ws_all ? ws_all + x.c * (kWsHead + a.R) : nullptr;
        obs = obs_all ? obs_all + x.c * a.R : nullptr;
        shift = 0.0;
    }

    __device__ __forceinline__ void emit(const KArgs &a, const LaneCtx &x, long r, double val)
    {
        if (a.discharge && x.live)
#if SMART_NT_STORE
            __builtin_nontemporal_store(val, &a.discharge[(x.c * a.R + r) * a.ld + x.n]);
#else
            a.discharge[(x.c * a.R + r) * a.ld + x.n] = val;
#endif
        if (want_obj) {
            const double e = obs[r];
            if (r == 0)
                shift = val;
            if (!is_nan_bits(e)) { // montecarlo.py:195-196
                const double d = val - e;
                const double u = val - shift;
                A += d;
                B += d * d;
                C1 += u;
                C2 += u * u;
                C3 += ws[kWsHead + r] * u;
            }
        }
    }

    // The same with the observation of report interval r and its deviation from the mean requested when interval
    // r - 1 was reported (prime() ahead of the first).  The run engine: -3.8 % (11.78 -> 11.33 ms on 6-hourly values).
    // NOT the step loops: there the request is caught by the next chunk's s_waitcnt lgkmcnt(0) (scalar loads return
    // out of order: there is no other wait) and the two live SGPR pairs cost more than they hide -- +2 % in the step
    // loop of the merged kernels, +5 % on the daily ensemble's literal chain (tools/gpu_r03_p.sh); the same request as
    // a vector load (counted by vmcnt, values in VGPRs) cost the step loop 3.5 % (tools/gpu_r03_q.sh).
    double e_nx = 0.0, w_nx = 0.0;

    __device__ __forceinline__ void prime(const KArgs &a, long r)
    {
        if (want_obj && a.R > 0) {
            const long nx = r < a.R ? r : a.R - 1;
            e_nx = obs[nx];
            w_nx = ws[kWsHead + nx];
        }
    }

    __device__ __forceinline__ void emit_ahead(const KArgs &a, const LaneCtx &x, long r, double val)
    {
        const double e = e_nx, w = w_nx;
        prime(a, r + 1);
        if (a.discharge && x.live)
#if SMART_NT_STORE
            __builtin_nontemporal_store(val, &a.discharge[(x.c * a.R + r) * a.ld + x.n]);
#else
            a.discharge[(x.c * a.R + r) * a.ld + x.n] = val;
#endif
        if (want_obj && r == 0)
            shift = val;
        if (want_obj && !is_nan_bits(e)) { // montecarlo.py:195-196
            const double d = val - e;
            const double u = val - shift;
            A += d;
            B += d * d;
            C1 += u;
            C2 += u * u;
            C3 += w * u;
        }
    }

    // The streamed step loop (FastModel::stream_stretch) reports inside its asm (smart_fast_arms.h: SMART_P_REPORT) -- what
    // emit() does, operation for operation, with three differences of form: the report's place in the discharge matrix is
    // this per-lane pointer, moved on by ld per report (begin_rows / next_row keep it in step when emit() reports); a
    // missing observation is told by the mark smart_obs_prepare left in its deviation; and there is no test for the lanes
    // beyond the batch -- they carry the batch's last sample (lane_ctx) and store what its own lane stores, where it does.
    double *row = nullptr;
    __device__ __forceinline__ void begin_rows(const KArgs &a, const LaneCtx &x, long r)
    {
        row = a.discharge ? a.discharge + (x.c * a.R + r) * a.ld + x.n : nullptr;
    }
    __device__ __forceinline__ void next_row(const KArgs &a)
    {
        if (a.discharge)
            row += a.ld;
    }

    // The same with the observation e = obs[r] and w = e - mean(e) requested ahead of time by the caller
    // (interval_loop_obs): no scalar-load latency between the last step of the interval and the moments.
    __device__ __forceinline__ void emit_prefetched(const KArgs &a, const LaneCtx &x, long r, double val, double e,
                                                    double w)
    {
        if (a.discharge && x.live)
            a.discharge[(x.c * a.R + r) * a.ld + x.n] = val;
        if (want_obj && r == 0)
            shift = val;
        if (want_obj && !is_nan_bits(e)) { // montecarlo.py:195-196
            const double d = val - e;
            const double u = val - shift;
            A += d;
            B += d * d;
            C1 += u;
            C2 += u * u;
            C3 += w * u;
        }
    }
};

template <class Model>
__device__ __forceinline__ void write_results(const KArgs &a, const LaneCtx &x, const Model &m, const Reporter &rep,
                                              double gw, const double *flows = nullptr)
{
    if (x.live)
        a.gw[x.c * a.N + x.n] = gw;
    if (rep.want_obj && x.live) {
        double o[8];
        finish_objectives(rep.ws, rep.A, rep.B, rep.C1, rep.C2, rep.C3, gw, a.gw_obs ? a.gw_obs[x.c] : quiet_nan(),
                          o);
        double *op = a.objfn + (x.c * a.N + x.n) * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            op[i] = o[i];
    }
    if (a.final_vars) { // (wave-uniform; a model that spreads a sample over a row collects it with DPP: all lanes in)
        double v[19];
        m.get_vars(v, flows);
        if (x.live) {
            double *fp = a.final_vars + (x.c * a.N + x.n) * 19;
#pragma unroll
            for (int i = 0; i < 19; ++i)
                fp[i] = v[i];
        }
    }
}

// The general launch body.  Model supplies: setup(area, dt, p), set_states(st12), excess(rain, peva),
// step(rain, peva, ex, acc, num, den), members q_out, q_in (sum of the five catchment outflows), q_gw (shallow +
// deep), get_vars(v19).
//
// forcing / obs / ws arrive as separate __restrict__ kernel parameters: only then can the compiler prove that
// the discharge stores do not clobber them and fetch the wave-uniform forcing with scalar loads (s_load)
// instead of one 64-lane vector load per step.
template <class Model, bool NP_MEAN>
__device__ __forceinline__ void run_ensemble(const KArgs &a, const double2 *__restrict__ forcing,
                                             const double *__restrict__ obs_all, const double *__restrict__ ws_all,
                                             double *lds, long block, long catchment, int sub = 0)
{
    const LaneCtx x = lanes_per_sample<Model>::value == 16 ? lane_ctx_rows(a, block, catchment, sub)
                                                           : lane_ctx(a, block, catchment);
    Model m;
    init_model(a, x, m);
    const double2 *__restrict__ f = forcing + x.c * a.T;

    // ---- warm-up over the first W steps of the same forcing; only the states survive (structure.py:118-121)
    double sink0 = 0.0, sink1 = 0.0, sink2 = 0.0; // warm-up: the sums are not needed
    time_loop(m, f, a.W, [&](const double2 v, const double ex) { m.step(v.x, v.y, ex, sink0, sink1, sink2); });

    // ---- the run proper (structure.py:143-146, 181-195)
    const bool summary = a.report_type == 1;
    const double inv_gap = 1.0 / (double)a.gap;
    Reporter rep;
    rep.init(a, x, obs_all, ws_all);
    // a wavefront of the row form is alone on its SIMD and sits out every latency it meets: the observation of a report
    // and its deviation are requested one report ahead (a report every step: one step of the model ahead)
    constexpr bool kAhead = lanes_per_sample<Model>::value > 1;
    if constexpr (kAhead)
        rep.prime(a, 0);
    const bool every_step = a.gap == 1; // (a mean over one value is the value: x / 1.0 == x, no division)

    // the last row of the storage table holds the seven outputs of the last step (structure.py:197): models that do not
    // track them take the last step on its own, after working them out from the state it starts from
    const bool capture = a.final_vars != nullptr && !Model::kTracksOutputs;
    if (every_step && !capture) {
        // A report every step (a daily run with daily reports: BASELINE config 2), either report type -- the mean over
        // one step is the step's outflow (x / 1.0 == x * 1.0 == x), and the groundwater sums of the reported rows are
        // those of all rows -- as a loop of its own: step, report, nothing to count and nothing to branch on.  In the
        // general loop below the report sits out of line behind `++k == len`: two taken branches and the counters'
        // bookkeeping per step cost a lone wavefront as much as half a step of the literal model (round 5: the
        // ill-conditioned rows of config 2 2.97 -> 2.18 ms, the stiff ones 1.62 -> see profiles/r05_config2.md).
        double num1 = 0.0, den1 = 0.0, total1 = 0.0;
        long r1 = 0;
        // Round 6: the lean report below -- built in round 5 for the row form alone -- for the one-sample-per-lane models
        // of this loop as well (stiff, guard, the literal rows' lane form): their lanes beyond the batch carry the last
        // sample (lane_ctx) and store its value to its address, the same bits.  Same-box A/B, every output bit equal
        // (profiles/r06_ab_lean_every.txt): smart_fast_stiff 1.30 -> 1.12 ms on config 2's stiff rows (21 scalar
        // instructions and 6 branches a step beside 85 vector ones before), smart_fast_illcond_lanes 4.12 -> 3.95 ms,
        // the daily ensemble of 1e6 samples 11.28 -> 10.97 ms.  SMART_LEAN_EVERY=0 builds the old form (A/B).
#ifndef SMART_LEAN_EVERY
#define SMART_LEAN_EVERY 1
#endif
        constexpr bool kLean = kAhead || (SMART_LEAN_EVERY && !Model::kBalanceSums);
        if constexpr (kLean) {
            // The row form's wavefront is alone on its SIMD: every instruction of the report is four to five cycles of
            // the step.  The PMC counters of config 2 (profiles/r05_config2.md) had 54 scalar instructions per wave-step
            // beside 152 vector ones; the report's share of them was the observation's index clamp and address (14), the
            // spilled kernel arguments read back for the matrix's row stride (8 v_readlane), the NaN test of a vector-loaded
            // observation (7), "is this report 0?" (5) and the EXEC mask of the store (4 + a branch).  Here: the
            // observation and its deviation through the constant address space (scalar loads, requested BEFORE the step
            // they belong to), two pointers that move on by 8 bytes, a missing observation told from the mark in its
            // deviation (one scalar compare, smart_obs_prepare), report 0 peeled off, and every lane of a row storing the
            // row's value (same address, same bits; the lanes beyond the batch carry the last sample's).
            const bool obj = rep.want_obj, store = a.discharge != nullptr;
            // (without observations the two loads still happen -- of the catchment's area, a double that is always there --
            // and nobody looks at them: a load costs the wavefront less than the branch around it)
            const_f64 pe = as_constant(obj ? rep.obs : a.area + x.c), pw = as_constant(obj ? rep.ws + kWsHead : a.area + x.c);
            const long inc = obj ? 1 : 0;
            const double zero = summary ? 0.0 : -0.0;
            double *row = store ? a.discharge + x.c * a.R * a.ld + x.n : nullptr;
            long stride; // the matrix's row stride in a VECTOR register pair: as a kernel argument it lives in a spilled scalar
                         // tuple, which hipcc reads back with eight v_readlane per step to get at these two
            asm volatile("v_mov_b64 %0, %1" : "=v"(stride) : "s"(a.ld));
            auto step_and_report = [&](const double2 v, const double ex, auto first) {
                const double e = *pe, w = *pw;
                pe += inc;
                pw += inc;
                double sink = 0.0;
                m.step(v.x, v.y, ex, sink, num1, den1); // (ex: the fast models' excess, worked out ahead by time_loop)
                // summary: the mean over one step is (0.0 + q) / 1; raw: q itself.  One addition for both: x + (-0.0) is x
                // for every x, the zeros and a NaN included
                const double val = m.q_out + zero;
                if (store) {
#if SMART_NT_STORE
                    __builtin_nontemporal_store(val, row);
#else
                    *row = val;
#endif
                    row += stride;
                }
                if (obj) {
                    if constexpr (decltype(first)::value)
                        rep.shift = val;
                    if (!is_missing_mark(w)) { // montecarlo.py:195-196
                        const double d = val - e;
                        const double u = val - rep.shift;
                        rep.A += d;
                        rep.B += d * d;
                        rep.C1 += u;
                        rep.C2 += u * u;
                        rep.C3 += w * u;
                    }
                }
            };
            if (a.T > 0) {
                time_loop(m, f, 1, [&](const double2 v, const double ex) { step_and_report(v, ex, std::true_type()); });
                time_loop(m, f + 1, a.T - 1,
                          [&](const double2 v, const double ex) { step_and_report(v, ex, std::false_type()); });
            }
            write_results(a, x, m, rep, num1 / den1, nullptr);
            return;
        }
        if constexpr (Model::kBalanceSums)
            m.begin_run();
        time_loop(m, f, a.T, [&](const double2 v, const double ex) {
            double acc1 = 0.0;
            m.step(v.x, v.y, ex, acc1, num1, den1);
            const double val = summary ? acc1 : m.q_out;
            rep.emit(a, x, r1, val);
            ++r1;
            if constexpr (Model::kBalanceSums)
                total1 += acc1;
        });
        if constexpr (Model::kBalanceSums)
            m.balance_sums(total1, num1, den1);
        write_results(a, x, m, rep, num1 / den1, nullptr);
        return;
    }

    double num = 0.0, den = 0.0;         // groundwater sums over every step (summary, structure.py:191)
    double num_raw = 0.0, den_raw = 0.0; // ... over the reported rows only (raw, structure.py:194-195)
    double acc = 0.0;                    // running sum of the current report interval
    double q_out_total = 0.0;            // ... and of all of them (models that derive the gw sums from balances)
    if constexpr (Model::kBalanceSums)
        m.begin_run();
    long k = 0, r = 0, len = a.first_len;
    auto one_step = [&](const double2 v, const double ex) {
        // the model adds this step's river outflow / groundwater outflow / total catchment outflow to the three
        // running sums itself, so that those additions sit in the same basic block as the step's dependent chains
        m.step(v.x, v.y, ex, acc, num, den);
        if (NP_MEAN)
            lds[k * kWave + x.lane] = m.q_out;
        if (__builtin_expect(++k == len, 0)) { // end of report interval r (wave-uniform, 1 step in `gap`)
            double val;
            if (summary) {
                if (NP_MEAN)
                    val = np_pairwise_lds(lds + x.lane, len) / (double)a.gap;
                else if (kAhead && every_step)
                    val = acc;
                else
                    val = Model::kExactDivide ? acc / (double)a.gap : acc * inv_gap;
            } else {
                val = m.q_out;
                num_raw += m.q_gw;
                den_raw += m.q_in;
            }
            if constexpr (kAhead)
                rep.emit_ahead(a, x, r, val);
            else
                rep.emit(a, x, r, val);
            ++r;
            k = 0;
            len = a.gap;
            if (Model::kBalanceSums)
                q_out_total += acc;
            acc = 0.0;
        }
    };
    double flows[7];
    time_loop(m, f, capture ? a.T - 1 : a.T, one_step);
    if (capture) {
        const double2 v = f[a.T - 1];
        m.flows_of_next_step(a.dt, v.x, v.y, flows);
        one_step(v, m.excess(v.x, v.y));
    }

    if constexpr (Model::kBalanceSums)
        m.balance_sums(q_out_total, num, den);
    write_results(a, x, m, rep, summary ? num / den : num_raw / den_raw, capture ? flows : nullptr);
}

// ---- code words of the pair blocks (smart_fast_arms.h: SMART_A_PAIRS_STRETCH) -----------------------------------------
// Per chunk of four steps two words, one per pair of steps: the byte offset of the pair's block from block 0.  Blocks
// lie kPairStride bytes apart, ordered by (chunk parity, pair, kind of the first step, kind of the second); kinds as the
// arms tell them apart on the bits of the forcing: rain != +0 -> rain step (2), else PE != +0 -> dry (1), else calm (0).
// A block that starts with a rain step is entered 4 bytes in.  Meaningful for sane forcing only (the QUICK waves).
constexpr long kPairStride = SMART_P_STRIDE, kPairStrideSplit = SMART_PS_STRIDE; // (smart_fast_arms.h; KArgs::pair_stride)
__host__ __device__ constexpr long code_chunks(long T) { return T / kChunk + 4; } // (+ the requests beyond a stretch)
__device__ __forceinline__ unsigned step_kind(const double2 v)
{
    return __builtin_bit_cast(unsigned long long, v.x) != 0 ? 2u : (__builtin_bit_cast(unsigned long long, v.y) != 0 ? 1u : 0u);
}
__device__ __forceinline__ unsigned pair_code(long chunk, int pair, unsigned k0, unsigned k1, long stride)
{
    return (unsigned)((((chunk & 1) * 2 + pair) * 9 + k0 * 3 + k1) * stride + (k0 == 2 ? 4 : 0));
}
// the every-step stream: pairs of steps, two register buffers by the pair's parity
constexpr long kEveryStride = SMART_E_STRIDE;
__host__ __device__ constexpr long every_pairs(long T) { return T / 2 + 4; } // (+ the requests beyond a stretch)
__device__ __forceinline__ unsigned every_code(long pair, unsigned k0, unsigned k1)
{
    return (unsigned)(((pair & 1) * 9 + k0 * 3 + k1) * kEveryStride + (k0 == 2 ? 4 : 0));
}
// ... for report gaps that are not whole chunks (SMART_A_GAP_STREAM): three variants of every block -- no report in the
// pair, the report behind its first arm, behind its second
__device__ __forceinline__ unsigned gap_code(long pair, long gap, unsigned k0, unsigned k1)
{
    const unsigned v = (2 * pair + 1) % gap == 0 ? 1u : ((2 * pair + 2) % gap == 0 ? 2u : 0u);
    return (unsigned)(((pair & 1) * 27 + v * 9 + k0 * 3 + k1) * kEveryStride + (k0 == 2 ? 4 : 0));
}
// the two code words of a chunk; four calm or four dry steps: one block for the chunk (36 + 2 x chunk parity + kind)
__device__ __forceinline__ uint2 chunk_codes(long chunk, unsigned k0, unsigned k1, unsigned k2, unsigned k3, long stride)
{
    if (k0 != 2 && k0 == k1 && k0 == k2 && k0 == k3)
        return make_uint2((unsigned)((36 + (chunk & 1) * 2 + k0) * stride), 0u);
    return make_uint2(pair_code(chunk, 0, k0, k1, stride), pair_code(chunk, 1, k2, k3, stride));
}

// ---- piecewise-constant forcing ------------------------------------------------------------------------------
// Is the forcing of every report interval one value repeated `gap` times?  (Hourly steps disaggregated from daily
// data: timeframe.py:167-186 splits each daily value equally over its 24 steps -- the reference's shipped example,
// its own regression test and the synthetic benchmark forcing all have this shape.)  Or at least constant over runs
// of k steps, k a divisor of the gap (6-hourly data in an hourly run with daily reports, the same pipeline)?  And is
// every value finite and >= +0 (the shortcuts of FastModel::step_arms)?  One int of flags per catchment:
//   bit i (i < n_div)   the forcing is NOT constant over the aligned runs of div[i] steps (div[0] = gap, descending)
//   kForcingInsane      some value is negative, -0, infinite or NaN
// A run of d steps is constant iff no step t with t % d != 0 differs from its predecessor: one compare per step, the
// remainders only where the forcing changes.  smart_forcing_scan answers for the whole launch (a.fflags); without a
// workspace every wavefront scans for itself (~0.1 ms for ten years of hourly steps).
__device__ __forceinline__ int forcing_flags_of_step(const KArgs &a, const double2 *__restrict__ f, long t)
{
    const double2 v = f[t];
    int bad = 0;
    const unsigned long long top = 0x7ff0000000000000ull; // sign clear and exponent below all ones <=> bits < top
    if (__builtin_bit_cast(unsigned long long, v.x) >= top || __builtin_bit_cast(unsigned long long, v.y) >= top)
        bad |= kForcingInsane;
    if (t > 0) {
        const double2 u = f[t - 1];
        if (!(same_bits(v.x, u.x) && same_bits(v.y, u.y))) {
            for (int i = 0; i < a.n_div; ++i)
                if (t % a.div[i] != 0)
                    bad |= 1 << i;
        }
    }
    return bad;
}

__device__ __forceinline__ int scan_forcing_wave(const KArgs &a, const double2 *__restrict__ f)
{
    int bad = 0;
    for (long t = threadIdx.x; t < a.T; t += kWave)
        bad |= forcing_flags_of_step(a, f, t);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        bad |= __shfl_xor(bad, off, kWave);
    return __builtin_amdgcn_readfirstlane(bad);
}

__device__ __forceinline__ int forcing_flags(const KArgs &a, const double2 *__restrict__ forcing, long c)
{
    return a.fflags ? a.fflags[c] : scan_forcing_wave(a, forcing + c * a.T);
}

// steps per run of constant forcing: the largest tested divisor of the gap that holds; 1 = the forcing varies
__device__ __forceinline__ long run_length(const KArgs &a, int flags)
{
    for (int i = 0; i < a.n_div; ++i)
        if (!((flags >> i) & 1))
            return a.div[i];
    return 1;
}

// forcing kinds of the merged summary kernels
constexpr int kForcingVarying = 0, kForcingIntervals = 1, kForcingRuns = 2;

__device__ __forceinline__ int forcing_kind(const KArgs &a, int flags)
{
    const long k = run_length(a, flags);
    return k == a.gap ? kForcingIntervals : (k >= 2 ? kForcingRuns : kForcingVarying);
}

// Walk the report intervals of a piecewise-constant forcing: one (rain, peva) pair per interval, fetched with scalar
// loads kGroup intervals ahead (a dry interval takes ~100 cycles, far less than a load's latency).
#ifndef SMART_IV_GROUP
#define SMART_IV_GROUP 4
#endif
constexpr int kGroup = SMART_IV_GROUP;

template <class Body>
__device__ __forceinline__ void interval_loop(const double2 *__restrict__ f, long i0, long i1, long gap, Body &&body)
{
    const long n_groups = (i1 - i0) / kGroup;
    double2 cur[kGroup], nxt[kGroup];
    if (n_groups > 0) {
#pragma unroll
        for (int j = 0; j < kGroup; ++j)
            cur[j] = f[(i0 + j) * gap];
    }
    for (long g = 0; g < n_groups; ++g) {
        const long pre = i0 + (g + 1 < n_groups ? g + 1 : g) * kGroup; // last group: harmless re-load of itself
#pragma unroll
        for (int j = 0; j < kGroup; ++j)
            nxt[j] = f[(pre + j) * gap];
#pragma unroll
        for (int j = 0; j < kGroup; ++j)
            body(i0 + g * kGroup + j, cur[j]);
#pragma unroll
        for (int j = 0; j < kGroup; ++j)
            cur[j] = nxt[j];
    }
    for (long i = i0 + n_groups * kGroup; i < i1; ++i)
        body(i, f[i * gap]);
}

// The same walk for the run proper: the observation of each interval and its deviation from the mean
// (Reporter::emit_prefetched) are requested one interval ahead of their use (an interval takes 700-7,000 cycles).
template <class Body>
__device__ __forceinline__ void interval_loop_obs(const double2 *__restrict__ f, const double *__restrict__ obs,
                                                  const double *__restrict__ dev, long i0, long i1, long gap,
                                                  Body &&body)
{
    if (i1 <= i0)
        return;
    double e = obs[i0], w = dev[i0];
    interval_loop(f, i0, i1, gap, [&](long i, const double2 v) {
        const long nx = i + 1 < i1 ? i + 1 : i;
        const double e_nx = obs[nx], w_nx = dev[nx];
        body(i, v, e, w);
        e = e_nx;
        w = w_nx;
    });
}

// ---- time-sliced launch ----------------------------------------------------------------------------------------
// A block of 64 samples is one wavefront for the whole time axis, so a launch of B blocks on S SIMDs lasts as long as
// the SIMDs that hold ceil(B / S) of them while the others idle (1e5 samples: 1,563 blocks on 1,024 SIMDs, 539 SIMDs
// with two).  Here the time axis of every block is cut into n_seg slices and each (block, slice) is its own
// workgroup, handed out as slots free up, so a SIMD that finishes early simply gets more slices.  Slice s of a block
// starts from the state slice s - 1 left in `seg_state` and waits for `seg_flag[slot] >= s`.
//
// Which (block, slice) a workgroup runs comes from a TICKET it draws when it starts (one atomic add on a counter in
// the workspace header), not from its workgroup id: ticket t is slice t / seg_blocks of block t % seg_blocks, so the
// slice a workgroup waits for (ticket t - seg_blocks) was drawn by a workgroup that is already running or done,
// whatever order the hardware dispatches workgroups in and however the ids map to XCDs -- forward progress needs
// nothing beyond "resident wavefronts keep executing".  The wait is bounded all the same (a preempted queue on a
// shared GPU): a slice that gives up raises kStatusSliceTimeout in the status word, writes NaN for its report rows
// and poisons its chain (negative flag), so that no number computed from a missing hand-over can pass for a result;
// the host reads the status word and repeats the launch unsliced (engine.py).  The arithmetic is that of the
// unsliced run, bit for bit.
#ifndef SMART_POLL_SLEEP
#define SMART_POLL_SLEEP 100
#endif
constexpr long kDefaultMaxPolls = 1000000; // x ~3 us = three seconds; a slice takes ~1 ms and waits for one predecessor

// the piece of work of this workgroup: block of 64 samples, catchment, time slice
struct Work {
    long block, c;
    int seg;
};

__device__ __forceinline__ Work claim_work(const KArgs &a, int sliced_kernel)
{
    Work w;
    if (a.n_seg > 1) {
        unsigned t = 0;
        if (threadIdx.x == 0)
            t = __hip_atomic_fetch_add(reinterpret_cast<unsigned *>(a.hdr) + kHdrTicket + sliced_kernel, 1u,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t = (unsigned)__builtin_amdgcn_readfirstlane((int)t);
        const long slot = (long)(t % (unsigned long)a.seg_blocks);
        w.seg = (int)(t / (unsigned long)a.seg_blocks);
        w.c = slot / a.n_blocks;
        w.block = slot % a.n_blocks;
    } else {
        w.block = blockIdx.x;
        w.c = blockIdx.y;
        w.seg = 0;
    }
    return w;
}

// true: the predecessor's hand-over is there; false: gave up, or the chain is already poisoned
__device__ __forceinline__ bool wait_for_slice(const KArgs &a, long slot, int seg)
{
    int *flag = a.seg_flag + slot;
    long polls = 0;
    int v;
    while ((v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) > -seg && v < seg) {
        __builtin_amdgcn_s_sleep(SMART_POLL_SLEEP);
        if (++polls > a.max_polls) {
            raise_status(a, kStatusSliceTimeout);
            return false;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return v >= seg;
}

// Producer side of the hand-over, in the order MI355X_MICROARCH.md prescribes for plain payload stores: this wave's
// stores drained (s_waitcnt vmcnt(0)) -> agent-scope release (buffer_wbl2 sc1: the XCD's L2 writes its dirty lines
// back) -> s_waitcnt vmcnt(0) AGAIN, as inline asm -> relaxed agent-scope flag store.  The fence's own wait is one
// hipcc (ROCm 7.2) drops when it can prove the wave's vmcnt scoreboard empty, and then the flag can overtake the
// write-back; inline asm is invisible to that pass.  tests/test_handover_isa.py disassembles every sliced kernel of
// the built library and checks that the wait is there (and the buffer_inv sc1 behind every poll).
__device__ __forceinline__ void publish_slice(const KArgs &a, long slot, int seg, bool good)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0)
        __hip_atomic_store(a.seg_flag + slot, good ? seg + 1 : -(seg + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Launch body of the merged regular variant for summary reports (Model::kIntervals), whole or time-sliced.
// Piecewise-constant forcing: the loop runs over report intervals, not steps -- one scalar load, one rain-excess
// evaluation and one wave-uniform wet / dry decision per interval, then
//   * dry lanes -> Model::dry_interval(): the routing half of the `gap` steps collapses to one 4 x 4 linear map (the
//                   reservoirs only drain: constant coefficients per sample), the soil half to one evaporation step
//                   with `gap` times the demand;
//   * wet lanes -> `gap` wet steps back to back, no compare and no branch per step.
// Any other forcing: the step loop of run_ensemble(), cut at the same report-interval boundaries.
// FORCING: kForcingIntervals, kForcingRuns (the same engine over runs of `run_len` steps, run_len a divisor of the gap:
// gap / run_len runs make a report interval, whose mean accumulates across them) or kForcingVarying.
// REPORT: what a report is (round 4 -- before, everything but the mean ran the general step loop of run_ensemble(),
// unsliced):
//   kReportMean   the mean of the interval's outflows (report='summary', gap >= 2; structure.py:189-191)
//   kReportLast   the outflow of the interval's last step (report='raw', gap >= 2, W and T multiples of the gap;
//                 :192-195), the groundwater ratio from the flows of those steps only.  Varying forcing: the arms with
//                 SMART_A_ROUTE_LAST, same instruction count as the summary loop.  Piecewise-constant forcing: the
//                 interval engine over n - 1 steps, a look at the reservoirs, the last step on its own.
//   kReportEvery  a report every step (gap 1, either report type: a mean over one value is the value, and raw rows are
//                 all rows): single-step arms with SMART_A_ROUTE_LAST and the report between them.
constexpr int kReportMean = 0, kReportLast = 1, kReportEvery = 2;

template <class Model, int FORCING, int REPORT = kReportMean>
__device__ __forceinline__ void run_ensemble_merged(const KArgs &a, const double2 *__restrict__ forcing,
                                                    const double *__restrict__ obs_all,
                                                    const double *__restrict__ ws_all, long block, long catchment,
                                                    int seg, int fflags)
{
    constexpr bool piecewise = FORCING != kForcingVarying; // a template parameter: the sides share no live value
    constexpr bool runs = FORCING == kForcingRuns;
    static_assert(REPORT == kReportMean || (!Model::kSplit && !runs), "raw / every-step reports: merged model, no runs");
    static_assert(REPORT != kReportEvery || !piecewise, "a report every step is a step loop");
    const LaneCtx x = lane_ctx(a, block, catchment);
    const long slot = catchment * a.n_blocks + block; // this block's place in seg_state / seg_flag
    Model m;
    init_model(a, x, m);
    const long gap = a.gap;
    const long run_len = runs ? run_length(a, fflags) : gap; // steps advanced at a time by the interval engine
    if constexpr (piecewise)
        m.setup_intervals(REPORT == kReportLast ? run_len - 1 : run_len); // (raw: the dry map stops ahead of the last step)
    const double2 *__restrict__ f = forcing + x.c * a.T;

    // A per-lane if / else: a wavefront whose lanes all fall on one side skips the other (s_cbranch_execz); in a mixed
    // wavefront each side runs under its lanes' mask.  Either way a lane's arithmetic depends on its own sample only.
    // Without rain the whole wavefront is on one side whatever its T values are (ex = -peva): the scalar unit sees that
    // from the forcing itself and spares the run its compare, its EXEC region and -- a calm run: neither rain nor
    // evaporation, the night block of sub-daily data -- the filling.  For forcing without negative or non-finite values
    // and waves with no layer above capacity (`quick`, as in the step loop); the kernels whose wet interval is the
    // asm loop.
    bool quick = false, fits = false;
    if constexpr (piecewise && Model::kWetAsm) {
        m.note_capacity();
        fits = m.over_mask == 0; // for good: only a caller's initial state puts a layer above its capacity (no GUARD here)
        quick = fits && !(fflags & kForcingInsane);
    }
    auto interval = [&](const double2 v, double &acc, double &num, double &den) {
        if constexpr (piecewise && Model::kWetAsm) {
            if (quick && __builtin_bit_cast(unsigned long long, v.x) == 0) {
                if (__builtin_bit_cast(unsigned long long, v.y) == 0) {
#if SMART_IV_DEFER
                    if (__builtin_amdgcn_ballot_w64(m.pend > 0.0) != 0)
                        m.flush_pending();
#endif
                    m.calm_interval(run_len, acc);
                } else {
                    m.dry_interval(-v.y, run_len, acc);
                }
                return;
            }
        }
        const double ex = m.excess(v.x, v.y);
        if (ex < 0.0) {
            m.dry_interval(ex, run_len, acc);
        } else {
#if SMART_IV_DEFER
            // what the lane's dry intervals since its last wet one have added to `pend` is taken from the layers now
            // (FastModel::dry_interval): one cascade per dry spell instead of one per dry interval
            if (m.pend > 0.0)
                m.flush_pending();
#endif
            if constexpr (piecewise && Model::kWetAsm)
                m.wet_interval(ex, run_len, acc, num, den, fits);
            else
                m.wet_interval(ex, run_len, acc, num, den);
        }
    };

    // raw reports: the interval as n - 1 steps, the flows the last step starts from, the last step
    [[maybe_unused]] auto interval_last = [&](const double2 v, double &qo, double &qg, double &qi) {
        if constexpr (piecewise && REPORT == kReportLast && Model::kWetAsm) {
            double sink = 0.0, n0 = 0.0, n1 = 0.0;
            if (quick && __builtin_bit_cast(unsigned long long, v.x) == 0) {
                if (__builtin_bit_cast(unsigned long long, v.y) == 0) {
                    m.calm_interval(gap - 1, sink);
                    m.last_step_flows(qo, qg, qi);
                    m.calm_interval(1, sink);
                } else {
                    m.dry_interval_last(-v.y, gap, qo, qg, qi);
                }
                return;
            }
            const double ex = m.excess(v.x, v.y);
            if (ex < 0.0) {
                m.dry_interval_last(ex, gap, qo, qg, qi);
            } else {
                m.wet_interval(ex, gap - 1, sink, n0, n1, fits);
                m.last_step_flows(qo, qg, qi);
                m.wet_interval(ex, 1, sink, n0, n1, fits);
            }
        }
    };

    // this workgroup's slice [g0, g1) of the W / gap warm-up intervals followed by the R report intervals
    // (n_seg == 1: everything).  Summary reports need W % gap == 0 (checked on the host, structure.py:190).
    const long n_warm = a.W / gap, n_all = n_warm + a.R;
    const long g0 = n_all * seg / a.n_seg, g1 = n_all * (seg + 1) / a.n_seg;
    const long wa = g0 < n_warm ? g0 : n_warm, wb = g1 < n_warm ? g1 : n_warm;
    const long ra = g0 > n_warm ? g0 - n_warm : 0, rb = g1 > n_warm ? g1 - n_warm : 0;
    const bool last = seg == a.n_seg - 1;

    Reporter rep;
    rep.init(a, x, obs_all, ws_all);
    // this catchment's observations and their deviations from the mean, as pointers of this function's own (the loops
    // that fetch them with scalar loads take them as __restrict__ parameters)
    [[maybe_unused]] const double *__restrict__ obs_c = obs_all ? obs_all + x.c * a.R : nullptr;
    [[maybe_unused]] const double *__restrict__ dev_c = ws_all ? ws_all + x.c * (kWsHead + a.R) + kWsHead : nullptr;
    [[maybe_unused]] const uint2 *codes_c = a.codes ? a.codes + x.c * code_chunks(a.T) : nullptr;
    // the same forcing for the asm that loads it itself (stream_stretch), NOT derived from the __restrict__ argument: a
    // pointer handed to an asm has escaped, and with `f` escaped every asm of the launch might have written the forcing
    // for all hipcc knows -- its own loads of it would no longer be scalar loads
    [[maybe_unused]] const double2 *f_asm = reinterpret_cast<const double2 *>(a.forcing) + x.c * a.T;
    double num = 0.0, den = 0.0, q_out_total = 0.0;
    [[maybe_unused]] double num_raw = 0.0, den_raw = 0.0; // raw reports: the two sums over the reported steps (structure.py:194-195)
    double *hand = a.seg_state + (slot * kSegFields) * kWave + x.lane;
    if (seg > 0) {
        if (!wait_for_slice(a, slot, seg)) {
            // the hand-over never came (or the chain is poisoned already): NaN for everything this slice owes, and
            // the same for its successors -- nothing computed from a missing state may pass for a result
            const double nan = quiet_nan();
            if (a.discharge && x.live)
                for (long r = ra; r < rb; ++r)
                    a.discharge[(x.c * a.R + r) * a.ld + x.n] = nan;
            if (last) { // the per-sample results, written as NaN bit patterns (no arithmetic: -fno-honor-nans)
                if (x.live) {
                    a.gw[x.c * a.N + x.n] = nan;
                    if (a.objfn)
                        for (int i = 0; i < 8; ++i)
                            a.objfn[(x.c * a.N + x.n) * 8 + i] = nan;
                    if (a.final_vars)
                        for (int i = 0; i < 19; ++i)
                            a.final_vars[(x.c * a.N + x.n) * 19 + i] = nan;
                }
            } else {
                publish_slice(a, slot, seg, false);
            }
            return;
        }
        m.load_state(hand, kWave);
        if constexpr (REPORT == kReportLast) {
            num_raw = hand[16 * kWave];
            den_raw = hand[13 * kWave]; // (fields 13, 14 carry the SPLIT models' extra states: free here)
        } else {
            q_out_total = hand[16 * kWave];
        }
        rep.A = hand[17 * kWave];
        rep.B = hand[18 * kWave];
        rep.C1 = hand[19 * kWave];
        rep.C2 = hand[20 * kWave];
        rep.C3 = hand[21 * kWave];
        rep.shift = hand[22 * kWave];
    }

    // Callers that ask for the final state vector (the SPLIT models) also get the seven outputs of the last step
    // (structure.py:197).  They are functions of the state the last STEP starts from, which an interval-at-a-time walk
    // never holds: the state at the start of the last report interval is parked in the sample's own final_vars row
    // (19 doubles, written for good at the very end), and once the run is over that interval is replayed step by step
    // -- gap - 1 steps and a look at the flows -- on a copy of the model.  Costs one interval per run, and nothing in
    // the kernels of callers that do not ask.
    // (the host launches a SPLIT kernel only for callers with a final_vars buffer)
    double *const park = Model::kSplit ? a.final_vars + (x.c * a.N + x.n) * 19 : nullptr;
    auto park_state = [&]() {
        if constexpr (Model::kSplit) {
            m.save_state(park, 1);
            if constexpr (!piecewise || SMART_IV_DEFER)
                park[15] = m.pend;
        }
    };

    // warm-up over the first W steps of the same forcing, only the states survive (structure.py:118-121); then the
    // run proper.  One branch around both loops: the flat side must not keep the interval coefficients alive.
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    const double inv_gap = 1.0 / (double)gap;
    const bool starts_run = ra == 0 && (rb > 0 || last);
    constexpr bool deferring = !piecewise || SMART_IV_DEFER; // is an evaporation demand carried in `pend`?
    if constexpr (piecewise && deferring)
        m.pend = seg > 0 ? hand[15 * kWave] : 0.0; // evaporation demand not yet taken from the layers (dry_interval)
    if constexpr (runs) {
        // `per` runs make a report interval; the walk is over runs, the report falls on every per-th of them
        const long per = gap / run_len;
        interval_loop(f, wa * per, wb * per, run_len, [&](long, const double2 v) { interval(v, s0, s1, s2); });
        if (starts_run)
            m.begin_run();
        long j = 0, r = ra;
        double acc = 0.0;
        rep.prime(a, ra);
        interval_loop(f, ra * per, rb * per, run_len, [&](long, const double2 v) {
            if (Model::kSplit && j == 0 && r == a.R - 1)
                park_state();
            interval(v, acc, num, den);
            if (++j == per) {
                rep.emit_ahead(a, x, r, acc * inv_gap);
                q_out_total += acc;
                acc = 0.0;
                j = 0;
                ++r;
            }
        });
    } else if constexpr (piecewise && REPORT == kReportLast) {
        interval_loop(f, wa, wb, gap, [&](long, const double2 v) { interval_last(v, s0, s1, s2); });
        if (starts_run)
            m.begin_run();
        if (rep.want_obj) {
            interval_loop_obs(f, rep.obs, rep.ws + kWsHead, ra, rb, gap,
                              [&](long r, const double2 v, const double e, const double w) {
                                  double qo, qg, qi;
                                  interval_last(v, qo, qg, qi);
                                  rep.emit_prefetched(a, x, r, qo, e, w);
                                  num_raw += qg;
                                  den_raw += qi;
                              });
        } else {
            interval_loop(f, ra, rb, gap, [&](long r, const double2 v) {
                double qo, qg, qi;
                interval_last(v, qo, qg, qi);
                rep.emit(a, x, r, qo);
                num_raw += qg;
                den_raw += qi;
            });
        }
    } else if constexpr (piecewise) {
        interval_loop(f, wa, wb, gap, [&](long, const double2 v) { interval(v, s0, s1, s2); });
        if (starts_run)
            m.begin_run();
        if (rep.want_obj) {
            interval_loop_obs(f, rep.obs, rep.ws + kWsHead, ra, rb, gap,
                              [&](long r, const double2 v, const double e, const double w) {
                                  if (Model::kSplit && r == a.R - 1)
                                      park_state();
                                  double acc = 0.0;
                                  interval(v, acc, num, den);
                                  rep.emit_prefetched(a, x, r, acc * inv_gap, e, w);
                                  q_out_total += acc;
                              });
        } else {
            interval_loop(f, ra, rb, gap, [&](long r, const double2 v) {
                if (Model::kSplit && r == a.R - 1)
                    park_state();
                double acc = 0.0;
                interval(v, acc, num, den);
                rep.emit(a, x, r, acc * inv_gap);
                q_out_total += acc;
            });
        }
    } else {
        // the step loop with deferred evaporation; the demand a slice has not yet taken from the layers travels in the
        // hand-over, so that a sliced run composes exactly like a whole one
        m.begin_lazy(seg > 0 ? hand[15 * kWave] : 0.0);
        long k = 0, r = ra;
        (void)k;
        double acc = 0.0;
        auto report = [&]() __attribute__((always_inline)) { // end of report interval r (wave-uniform)
            auto emit = [&](const double val) __attribute__((always_inline)) {
                rep.emit(a, x, r, val);
                rep.next_row(a);
            };
            if constexpr (REPORT == kReportMean) {
                emit(acc * inv_gap);
                ++r;
                k = 0;
                q_out_total += acc;
                acc = 0.0;
                if (Model::kSplit && r == a.R - 1)
                    park_state();
            } else if constexpr (REPORT == kReportLast) { // acc, q_gw, q_in: what SMART_A_ROUTE_LAST left of the last step
                emit(acc);
                ++r;
                num_raw += m.q_gw;
                den_raw += m.q_in;
            }
        };
        // a report every step: e, w = the observation of this report and its deviation from the mean, fetched with the
        // forcing (time_loop_arms_each); the groundwater ratio from the balances, as for the means.  The discharge row
        // of a report is one uniform pointer that moves on by ld, the lane adds its own place.
        // (store or not, observations or not: decided once per launch, outside the loop -- four instances of it)
        [[maybe_unused]] double *row = a.discharge ? a.discharge + (x.c * a.R + ra) * a.ld : nullptr;
        [[maybe_unused]] auto report_every = [&](auto store_tag, auto obs_tag, const double e, const double w) {
            const double val = acc; // (the outflow of the step just taken: SMART_A_ROUTE_LAST)
            if constexpr (decltype(store_tag)::value) {
                if (x.live)
                    row[x.n] = val;
                row += a.ld;
            }
            if (decltype(obs_tag)::value && !is_missing_mark(w)) { // montecarlo.py:195-196 (e is a number: smart_obs_prepare)
                const double d = val - e;
                const double u = val - rep.shift;
                rep.A += d;
                rep.B += d * d;
                rep.C1 += u;
                rep.C2 += u * u;
                rep.C3 += w * u;
            }
            q_out_total += val;
        };
#if SMART_STEP_ARMS
        // The three asm arms of FastModel::step_arms.  The shortcuts of the dry and the calm arm need forcing without
        // negative or non-finite values (smart_forcing_scan) and no layer above capacity: `quick`.
        // ONE instance of the loop serves the warm-up and the run: the walk is over two stretches of whole report
        // intervals (three inlined copies of the loop -- warm-up, run, run with a test per step -- cost 30 VGPRs at
        // their joins).
        const bool quick = m.zero_ok && !(fflags & kForcingInsane);
        auto walk = [&](auto quick_tag) __attribute__((always_inline)) {
            constexpr bool Q = decltype(quick_tag)::value;
#pragma nounroll
            for (int stretch = 0; stretch < 2; ++stretch) { // the warm-up intervals, then the report intervals
                const long i0 = stretch ? ra : wa, i1 = stretch ? rb : wb;
                if (stretch == 1) {
                    rep.begin_rows(a, x, ra);
                    acc = 0.0; // (the warm-up's sum)
                    if (starts_run)
                        m.begin_run();
                    if (Model::kSplit && ra == a.R - 1 && rb > ra)
                        park_state();
                }
                if constexpr (REPORT == kReportEvery) { // (gap == 1: intervals are steps)
                    if (stretch == 0) {
                        time_loop_arms<Q, true>(m, f + i0, i1 - i0, acc, [] {});
                    } else {
                        auto run_steps = [&](auto store_tag, auto obs_tag) {
                            constexpr bool OBS = decltype(obs_tag)::value;
                            long first = i0;
                            if (i0 == 0 && i1 > 0) { // report 0 on its own: it sets the constant the moments are taken about
                                m.template step_arms<Q, true>(f[0], acc);
                                rep.shift = acc;
                                report_every(store_tag, obs_tag, OBS ? obs_c[0] : 0.0, OBS ? dev_c[0] : 0.0);
                                first = 1;
                            }
                            auto each = [&](long from, long n) __attribute__((always_inline)) {
                                time_loop_arms_each<Q, OBS>(m, f + from, OBS ? obs_c + from : nullptr,
                                                            OBS ? dev_c + from : nullptr, n, acc,
                                                            [&](long, const double e, const double w) {
                                                                report_every(store_tag, obs_tag, e, w);
                                                            });
                            };
#if SMART_PAIR_BLOCKS
                            // whole groups of four steps, from a multiple of four on: the stream of SMART_A_EVERY_STREAM
                            if constexpr (Q && !Model::kSplit) {
                                const unsigned pc_lo = (unsigned)__builtin_amdgcn_s_getpc(); // (see arm_intervals)
                                if (a.estream && pc_lo > 0x00400000u && pc_lo < 0xffc00000u) {
                                    constexpr bool STORE = decltype(store_tag)::value;
                                    const long lead = (4 - first % 4) % 4 < i1 - first ? (4 - first % 4) % 4 : i1 - first;
                                    each(first, lead);
                                    first += lead;
                                    const long quads = (i1 - first) / 4;
                                    if (quads > 0) {
                                        double *lane_row = STORE ? row + x.n : nullptr;
                                        m.template stream_every<STORE, OBS>(
                                            a.estream + (x.c * every_pairs(a.T) + first / 2) * 8,
                                            a.ecodes + x.c * every_pairs(a.T) + first / 2, (int)quads, acc, rep.A, rep.B,
                                            rep.C1, rep.C2, rep.C3, rep.shift, q_out_total, lane_row, a.ld);
                                        if constexpr (STORE)
                                            row += quads * 4 * a.ld;
                                        first += quads * 4;
                                    }
                                }
                            }
#endif
                            each(first, i1 - first);
                        };
                        if (row && rep.want_obj)
                            run_steps(std::true_type{}, std::true_type{});
                        else if (row)
                            run_steps(std::true_type{}, std::false_type{});
                        else if (rep.want_obj)
                            run_steps(std::false_type{}, std::true_type{});
                        else
                            run_steps(std::false_type{}, std::false_type{});
                    }
                } else {
                    long done = 0;
#if SMART_PAIR_BLOCKS
                    // The pair blocks (FastModel::stream_stretch: the whole stretch, reports included, in one asm): report
                    // gaps of whole pairs of chunks, the kinds of the steps from smart_forcing_scan's code words.  The last
                    // interval of the forcing array stays with the loop below (the asm requests two chunks ahead).  Its
                    // jumps add a 32-bit offset to the address of block 0 without a carry: not for a code object that
                    // straddles a 4 GB line -- that launch walks the threaded chunks.  (Observations and deviations
                    // through a.obs / a.ws, not through the __restrict__ arguments: see f_asm.)
                    // Gaps that are not whole chunks (2, 3, 6 ... steps; FastModel::stream_gap): the stream of records, the
                    // report behind the arms whose steps end an interval.  From an interval that starts on a multiple of
                    // four steps on, as many groups of lcm(gap, 4) steps as the stretch has.
                    if constexpr (Q && !Model::kSplit) {
                        const unsigned pc_lo = (unsigned)__builtin_amdgcn_s_getpc();
                        if (a.estream && gap % kChunk != 0 && pc_lo > 0x00400000u && pc_lo < 0xffc00000u) {
                            const long unit = gap % 2 ? 4 : 2; // intervals per lcm(gap, 4) steps
                            const long lead = (unit - i0 % unit) % unit;
                            const long n_stream = i1 - i0 > lead ? (i1 - i0 - lead) / unit * unit : 0;
                            if (n_stream > 0) {
                                const bool reporting = stretch == 1;
                                if (lead > 0)
                                    arm_intervals<Q, REPORT == kReportLast>(m, f + i0 * gap, lead, gap, false, acc,
                                                                            [&]() __attribute__((always_inline)) {
                                                                                if (stretch == 1)
                                                                                    report();
                                                                            });
                                const long first = (i0 + lead) * gap;
                                double unused = 0.0;
                                m.template stream_gap<REPORT == kReportLast>(
                                    a.estream + (x.c * every_pairs(a.T) + first / 2) * 8,
                                    a.ecodes + x.c * every_pairs(a.T) + first / 2, (int)(n_stream * gap / 4), reporting,
                                    a.discharge != nullptr, rep.want_obj, r == 0, inv_gap, acc, rep.A, rep.B, rep.C1, rep.C2,
                                    rep.C3, rep.shift, REPORT == kReportLast ? num_raw : q_out_total,
                                    REPORT == kReportLast ? den_raw : unused, rep.row, a.ld);
                                if (reporting)
                                    r += n_stream;
                                done = lead + n_stream;
                            }
                        }
                    }
                    if constexpr (Q) {
                        const unsigned pc_lo = (unsigned)__builtin_amdgcn_s_getpc();
                        // whole intervals whose last chunk has two chunks of the array behind it (the asm's requests) --
                        // and, for the models that park their state ahead of the run's last report interval, not that one
                        const long cpi = gap / kChunk, room = gap % kChunk == 0 ? (a.T / kChunk - 2) / cpi - i0 : 0;
                        long n_stream = room < i1 - i0 ? room : i1 - i0;
                        if (Model::kSplit && stretch == 1 && n_stream > a.R - 1 - r)
                            n_stream = a.R - 1 - r;
                        if (codes_c && n_stream > 0 && pc_lo > 0x00400000u && pc_lo < 0xffc00000u) {
                            const bool reporting = stretch == 1, fetch = reporting && rep.want_obj;
                            double unused = 0.0;
                            auto go = [&](auto odd_tag) __attribute__((always_inline)) {
                                constexpr bool ODD = decltype(odd_tag)::value;
                                m.template stream_stretch<REPORT == kReportLast, ODD>(
                                    f_asm + i0 * gap, codes_c + i0 * cpi, fetch ? a.obs + x.c * a.R + i0 : nullptr,
                                    fetch ? a.ws + x.c * (kWsHead + a.R) + kWsHead + i0 : nullptr, n_stream,
                                    (int)(ODD ? cpi : cpi / 2), ((i0 * cpi) & 1) != 0, reporting, a.discharge != nullptr,
                                    r == 0, inv_gap, acc, rep.A, rep.B, rep.C1, rep.C2, rep.C3, rep.shift,
                                    REPORT == kReportLast ? num_raw : q_out_total,
                                    REPORT == kReportLast ? den_raw : unused, rep.row, a.ld);
                            };
                            if (cpi & 1) // (gaps of an odd number of chunks -- 4, 12, 20 steps ...: the form that counts
                                go(std::true_type{}); // chunks in both buffers' tails)
                            else
                                go(std::false_type{});
                            if (reporting) {
                                r += n_stream;
                                if (Model::kSplit && r == a.R - 1)
                                    park_state();
                            }
                            done = n_stream;
                        }
                    }
#endif
                    arm_intervals<Q, REPORT == kReportLast>(m, f + (i0 + done) * gap, i1 - i0 - done, gap, i1 * gap == a.T,
                                                            acc, [&]() __attribute__((always_inline)) {
                                                                if (stretch == 1)
                                                                    report();
                                                            });
                }
            }
        };
        if (quick)
            walk(std::true_type{});
        else
            walk(std::false_type{});
#else
        // no rain and no evaporation in this step (forcing is wave-uniform: scalar unit); never when a layer may be
        // above its capacity (Model::zero_ok)
        const unsigned long long not_ok = m.zero_ok ? 0ull : ~0ull;
        auto calm = [not_ok](const double2 v) {
            return (__builtin_bit_cast(unsigned long long, v.x) | __builtin_bit_cast(unsigned long long, v.y) | not_ok) == 0;
        };
        time_loop_lazy(m, f + wa * gap, (wb - wa) * gap, calm, s0, s1, s2, [] {});
        if (starts_run)
            m.begin_run();
        if (Model::kSplit && ra == a.R - 1 && rb > ra)
            park_state();
        if (gap % kChunk == 0) { // intervals end on chunk boundaries: one test per chunk of steps, not per step
            time_loop_lazy(m, f + ra * gap, (rb - ra) * gap, calm, acc, num, den, [&]() {
                k += kChunk;
                if (__builtin_expect(k == gap, 0))
                    report();
            });
        } else { // the test after every step
            const double2 *__restrict__ fr = f + ra * gap;
            for (long t = 0; t < (rb - ra) * gap; ++t) {
                const double2 v = fr[t];
                const double ex = m.excess(v.x, v.y);
                m.step_lazy(ex, __builtin_amdgcn_ballot_w64(ex >= 0.0), calm(v), acc, num, den);
                if (__builtin_expect(++k == gap, 0))
                    report();
            }
        }
#endif
    }
    if constexpr (deferring) {
        if (last)
            m.flush_pending(); // the final state vector wants the layers as the reference leaves them
    }

    if (last) {
        m.balance_sums(q_out_total, num, den);
        double gw = REPORT == kReportLast ? num_raw / den_raw : num / den;
        if constexpr (Model::kSplit) { // replay the last report interval from the parked state, step by step
            double flows[7];
            Model m0 = m;
            m0.load_state(park, 1);
            if constexpr (deferring) {
                m0.begin_lazy(park[15]);
                m0.flush_pending();
            }
            const double2 *__restrict__ fl = f + (a.R - 1) * gap;
            double t0 = 0.0, t1 = 0.0, t2 = 0.0;
            for (long j = 0; j + 1 < gap; ++j) {
                const double2 v = fl[j];
                m0.step(v.x, v.y, m0.excess(v.x, v.y), t0, t1, t2);
            }
            const double2 v = fl[gap - 1];
            m0.flows_of_next_step(a.dt, v.x, v.y, flows);
            write_results(a, x, m, rep, gw, flows);
        } else {
            write_results(a, x, m, rep, gw);
        }
    } else {
        m.save_state(hand, kWave);
        if constexpr (REPORT == kReportLast) {
            hand[16 * kWave] = num_raw;
            hand[13 * kWave] = den_raw;
        } else {
            hand[16 * kWave] = q_out_total;
        }
        hand[17 * kWave] = rep.A;
        hand[18 * kWave] = rep.B;
        hand[19 * kWave] = rep.C1;
        hand[20 * kWave] = rep.C2;
        hand[21 * kWave] = rep.C3;
        hand[22 * kWave] = rep.shift;
        if constexpr (deferring)
            hand[15 * kWave] = m.pend;
        if (!(a.debug_drop && slot == 0 && seg == 0)) // test knob: a hand-over that never arrives
            publish_slice(a, slot, seg, true);
    }
}

} // namespace smart
