// smart_capi.hip -- the C ABI of include/smart_amd.h: validation, launch plumbing, and the two small
// kernels around the ensemble launch (observation statistics, objective functions of a stored matrix).
#include "../../include/smart_amd.h"
#include "smart_fast_entry.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

namespace smart {
void launch_literal(const KArgs &a, dim3 grid, size_t lds_bytes, hipStream_t s, bool rows);
void launch_onestep(long n, const double *in, double *out, hipStream_t s);
void launch_river(long n, const double *in, double *out, hipStream_t s);

static_assert(kStatusSliceTimeout == SMART_STATUS_SLICE_TIMEOUT && kStatusStalePlan == SMART_STATUS_STALE_PLAN,
              "status bits of smart_device.h and include/smart_amd.h");

static thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) // also used by smart_hostio.cpp
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

static int hip_fail(hipError_t e, const char *what)
{
    return fail(SMART_E_NO_DEVICE, "%s: %s", what, hipGetErrorString(e));
}

#define HIP_TRY(expr)                                                                                                  \
    do {                                                                                                               \
        hipError_t _e = (expr);                                                                                        \
        if (_e != hipSuccess)                                                                                          \
            return hip_fail(_e, #expr);                                                                                \
    } while (0)

// ---- block reduction helper (256 threads), deterministic order -----------------------------------------
__device__ inline double block_sum(double v, double *sh)
{
    const int tid = threadIdx.x;
    sh[tid] = v;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if (tid < s)
            sh[tid] += sh[tid + s];
        __syncthreads();
    }
    const double r = sh[0];
    __syncthreads();
    return r;
}

// statistics of one observation series (NaN = missing, montecarlo.py:195-196): st[0..4] = n, mean, sum,
// sum((e-mean)^2), sum(e-mean); dev[r] = e[r] - mean when dev != null -- and kMissingObs where e[r] is missing: a NaN of a
// payload no arithmetic produces, so that a kernel with a report every step can tell a missing observation by ONE 32-bit
// scalar compare on the deviation it loads anyway (Reporter's other users test e itself and never read dev then)
__device__ inline void obs_stats(const double *obs, long R, double *st, double *dev, double *sh)
{
    double cnt = 0.0, s = 0.0;
    for (long r = threadIdx.x; r < R; r += blockDim.x) {
        const double e = obs[r];
        if (!is_nan_bits(e)) {
            cnt += 1.0;
            s += e;
        }
    }
    cnt = block_sum(cnt, sh);
    s = block_sum(s, sh);
    const double mean = s / cnt;
    double s2 = 0.0, s1 = 0.0;
    for (long r = threadIdx.x; r < R; r += blockDim.x) {
        const double e = obs[r];
        const bool missing = is_nan_bits(e);
        const double d = !missing ? e - mean : 0.0;
        s2 += d * d;
        s1 += d;
        if (dev)
            dev[r] = missing ? __builtin_bit_cast(double, kMissingObs) : d;
    }
    s2 = block_sum(s2, sh);
    s1 = block_sum(s1, sh);
    st[0] = cnt;
    st[1] = mean;
    st[2] = s;
    st[3] = s2;
    st[4] = s1;
}

__global__ __launch_bounds__(256) void smart_obs_prepare(const double *obs, long R, double *ws)
{
    __shared__ double sh[512];
    __shared__ double st[5];
    const long c = blockIdx.x;
    double *w = ws + c * (kWsHead + R);
    obs_stats(obs + c * R, R, st, w + kWsHead, sh);
    if (threadIdx.x < 5)
        w[threadIdx.x] = st[threadIdx.x];
}

// Objective functions of a stored discharge matrix sim[R][ld] (sample-minor).  HBM-bound: the matrix is read
// exactly once, 8 * R bytes per sample, every wavefront load one contiguous 512-byte row segment, UNROLL of them in
// flight per lane.  Moments are taken about the observation mean (the same one-pass form as the fused path of the
// time-loop kernel).  A workgroup is WX wavefronts wide along the samples and WR deep along the report rows:
//   WX = 4, WR = 1 : one lane walks all rows of its sample (large N: enough wavefronts, 2 KB contiguous per row);
//   WX = 1, WR = 8 : 8 wavefronts share 64 samples and take the rows round-robin, partial moments are reduced
//                    through LDS in a fixed order (N ~ 1e5: 8x more wavefronts in flight).
template <int WX, int WR, int UNROLL>
__global__ __launch_bounds__(WX *WR *kWave) void smart_objfn_matrix(long N, long R, const double *__restrict__ sim,
                                                                    long ld, const double *__restrict__ obs,
                                                                    const double *__restrict__ gw_sim, double gw_obs,
                                                                    double *__restrict__ objfn)
{
    __shared__ double sh[512];
    __shared__ double st[5];
    __shared__ double part[WR > 1 ? WR : 1][5][kWave];
    obs_stats(obs, R, st, nullptr, sh); // every thread stores the same five values to st
    __syncthreads();
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int wx = wave % WX, wr = wave / WX;
    long n = ((long)blockIdx.x * WX + wx) * kWave + lane;
    const bool live = n < N;
    if (!live)
        n = N - 1;
    const double ebar = st[1];
    const double *col = sim + n;
    const double shift = col[0]; // any constant per sample works (finish_objectives); the first value keeps the digits
    double A = 0.0, B = 0.0, C1 = 0.0, C2 = 0.0, C3 = 0.0;
    auto add = [&](double e, double s) {
        if (!is_nan_bits(e)) { // montecarlo.py:195-196
            const double d = s - e, u = s - shift;
            A += d;
            B += d * d;
            C1 += u;
            C2 += u * u;
            C3 += (e - ebar) * u;
        }
    };
    long r = wr;
    for (; r + (UNROLL - 1) * WR < R; r += UNROLL * WR) { // UNROLL independent row loads in flight
        double s[UNROLL];
#pragma unroll
        for (int j = 0; j < UNROLL; ++j)
            s[j] = col[(r + j * WR) * ld];
#pragma unroll
        for (int j = 0; j < UNROLL; ++j)
            add(obs[r + j * WR], s[j]);
    }
    for (; r < R; r += WR)
        add(obs[r], col[r * ld]);
    double m[5] = {A, B, C1, C2, C3};
    if (WR > 1) {
#pragma unroll
        for (int k = 0; k < 5; ++k)
            part[wr][k][lane] = m[k];
        __syncthreads();
        if (wr != 0)
            return;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            double t = part[0][k][lane];
#pragma unroll
            for (int w = 1; w < WR; ++w)
                t += part[w][k][lane];
            m[k] = t;
        }
    }
    if (live) {
        double o[8];
        finish_objectives(st, m[0], m[1], m[2], m[3], m[4], gw_sim ? gw_sim[n] : 0.0,
                          gw_sim ? gw_obs : __builtin_nan(""), o);
        double *op = objfn + n * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            op[k] = o[k];
    }
}

// ---- small kernels around the launch -------------------------------------------------------------------------
// What kind of forcing (constant over the report interval, over shorter runs, varying; sane values) is answered once
// per catchment, by the whole chip, before the ensemble kernels start.  fflags[c] is zeroed by smart_workspace_reset;
// see forcing_flags_of_step (smart_device.h) for the bits.
__global__ void smart_forcing_scan(KArgs a, const double2 *__restrict__ forcing)
{
    const double2 *__restrict__ f = forcing + (long)blockIdx.y * a.T;
    int bad = 0;
    bool wild = false; // a NaN or an infinity (exponent bits all ones)
    const unsigned long long top = 0x7ff0000000000000ull;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < a.T; t += (long)gridDim.x * blockDim.x) {
        bad |= forcing_flags_of_step(a, f, t);
        const double2 v = f[t];
        wild = wild || (__builtin_bit_cast(unsigned long long, v.x) & top) == top ||
               (__builtin_bit_cast(unsigned long long, v.y) & top) == top;
    }
    // the kinds of the steps, chunk by chunk, as the code words of the pair blocks (smart_device.h: pair_code); the
    // entries beyond the last whole chunk are requested (two chunks ahead) but never acted on: block 0's
    if (a.codes) {
        uint2 *codes = const_cast<uint2 *>(a.codes) + (long)blockIdx.y * code_chunks(a.T);
        for (long ch = (long)blockIdx.x * blockDim.x + threadIdx.x; ch < code_chunks(a.T);
             ch += (long)gridDim.x * blockDim.x) {
            uint2 w = make_uint2(0u, 0u);
            if ((ch + 1) * kChunk <= a.T) {
                const double2 *__restrict__ v = f + ch * kChunk;
                w = chunk_codes(ch, step_kind(v[0]), step_kind(v[1]), step_kind(v[2]), step_kind(v[3]), a.pair_stride);
            }
            codes[ch] = w;
        }
    }
    // a report every step: the run as records of (rain, PE, observation, deviation) and a code word per pair of steps
    if (a.estream) {
        double *rec = const_cast<double *>(a.estream) + (long)blockIdx.y * every_pairs(a.T) * 8;
        unsigned *codes = const_cast<unsigned *>(a.ecodes) + (long)blockIdx.y * every_pairs(a.T);
        const double *obs = a.obs ? a.obs + (long)blockIdx.y * a.R : nullptr;
        const double *dev = a.ws ? a.ws + (long)blockIdx.y * (kWsHead + a.R) + kWsHead : nullptr;
        for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < every_pairs(a.T); p += (long)gridDim.x * blockDim.x) {
            double r[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            unsigned code = 0;
            if (2 * p + 1 < a.T) {
                for (int j = 0; j < 2; ++j) { // the report that step t ends: number (t + 1) / gap - 1, if any
                    const long t = 2 * p + j, rr = (t + 1) % a.gap == 0 ? (t + 1) / a.gap - 1 : -1;
                    r[4 * j] = f[t].x;
                    r[4 * j + 1] = f[t].y;
                    r[4 * j + 2] = obs && dev && rr >= 0 && rr < a.R ? obs[rr] : 0.0;
                    r[4 * j + 3] = obs && dev && rr >= 0 && rr < a.R ? dev[rr] : 0.0;
                }
                code = a.gap == 1 ? every_code(p, step_kind(f[2 * p]), step_kind(f[2 * p + 1]))
                                  : gap_code(p, a.gap, step_kind(f[2 * p]), step_kind(f[2 * p + 1]));
            }
            for (int j = 0; j < 8; ++j)
                rec[p * 8 + j] = r[j];
            codes[p] = code;
        }
    }
    if (bad)
        __hip_atomic_fetch_or(a.fflags + blockIdx.y, bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // a NaN or an infinity in the forcing: the fast kernels are not made for it (what the reference's branches do with
    // it only the literal kernel reproduces) -- the launch goes on, and says so in its status word
    if (wild && a.hdr)
        __hip_atomic_fetch_or(a.hdr + kHdrStatus, kStatusNonFiniteForcing, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Zeroes the header (status word, tickets), the forcing flags and the slice counters.  A kernel rather than
// hipMemsetAsync: inside a captured HIP graph the memset node of ROCm 7.2 was not ordered before the ensemble kernel
// (replays started with the counters of the previous replay; tools/debug/graph_dbg.py), kernel nodes are.
__global__ void smart_workspace_reset(int *hdr, long n_hdr, int *flags, long n_flags)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_hdr)
        hdr[i] = 0;
    if (i < n_flags)
        flags[i] = 0;
}

// smart_plan_ensemble: the arithmetic classes present among the blocks of 64 rows, and the kinds of forcing
constexpr int kHdrPlan = 8, kHdrPlanIllCond = 9;

__global__ __launch_bounds__(kWave) void smart_classify_rows(KArgs a)
{
    const int cls = wave_class(a, (long)blockIdx.x, (long)blockIdx.y);
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_or(a.hdr + kHdrPlan, 1 << cls, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // round 6: HOW MANY blocks take the literal arithmetic -- the launch picks that kernel's form from it
        if (cls == 3)
            __hip_atomic_fetch_add(a.hdr + kHdrPlanIllCond, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ void smart_classify_forcing(KArgs a, int *hdr)
{
    const long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < a.n_catch) {
        const int kind = forcing_kind(a, a.fflags[c]);
        __hip_atomic_fetch_or(hdr + kHdrPlan,
                              kind == kForcingIntervals ? SMART_PLAN_FORCING_PIECEWISE
                                                        : (kind == kForcingRuns ? SMART_PLAN_FORCING_RUNS
                                                                                : SMART_PLAN_FORCING_VARYING),
                              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- per-device context: what the library caches about a device, and the streams a multi-kernel launch forks onto
// A call launches at most kMaxTodo kernels side by side: up to three kernels of the regular rows (one per kind of
// forcing among the catchments: piecewise, runs, varying) + the stiff, guard and ill-conditioned classes = 6; the first
// runs on the caller's stream, each other one on an auxiliary stream of its own.
constexpr int kMaxDevices = 64, kMaxTodo = 8, kMaxAux = kMaxTodo - 1;

struct DeviceCtx {
    std::once_flag once;
    int n_simd = 0;
    std::mutex mu; // guards everything below
    hipStream_t aux[kMaxAux] = {};
    hipEvent_t fork = nullptr, join[kMaxAux] = {};
    size_t lds_size[kNumFastKernels][17] = {};
    bool lds_known[kNumFastKernels][17] = {};
};

static DeviceCtx g_dev[kMaxDevices];

static DeviceCtx *device_ctx()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices)
        return nullptr;
    DeviceCtx *d = &g_dev[dev];
    std::call_once(d->once, [d, dev] {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
            d->n_simd = cus * 4;
    });
    return d->n_simd > 0 ? d : nullptr;
}

static const void *fast_kernel(FastKernel k)
{
    if (const void *f = fast_kernel_intervals(k))
        return f;
    if (const void *f = fast_kernel_steps(k))
        return f;
    if (const void *f = fast_kernel_runs(k))
        return f;
    if (const void *f = fast_kernel_reports(k))
        return f;
    return fast_kernel_guarded(k);
}

static const char *const kFastKernelNames[kNumFastKernels] = {
    "smart_fast_intervals_exits", "smart_fast_intervals", "smart_fast_intervals_states", "smart_fast_steps",
    "smart_fast_steps_states", "smart_fast_plain", "smart_fast_stiff", "smart_fast_guard", "smart_fast_illcond",
    "smart_fast_runs_exits", "smart_fast_runs", "smart_fast_runs_states", "smart_fast_steps_raw",
    "smart_fast_intervals_raw", "smart_fast_steps_every", "smart_fast_illcond_lanes"};

// dynamic LDS that lets exactly `per_cu` workgroups of kernel k be resident on a CU (0: no such size); d->mu held
static size_t lds_for_residency(DeviceCtx *d, FastKernel k, int per_cu)
{
    if (per_cu < 1 || per_cu > 16)
        return 0;
    if (!d->lds_known[k][per_cu]) {
        size_t found = 0;
        for (size_t x = (size_t)(160 * 1024 / per_cu) / 256 * 256;
             x >= 1024 && x > (size_t)(160 * 1024 / (per_cu + 1)) - 2048; x -= 256) {
            int nb = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fast_kernel(k), kWave, x) != hipSuccess)
                break;
            if (nb == per_cu) {
                found = x;
                break;
            }
            if (nb > per_cu)
                break;
        }
        d->lds_size[k][per_cu] = found;
        d->lds_known[k][per_cu] = true;
    }
    return d->lds_size[k][per_cu];
}

// ---- workspace layout: header | [C][8 + R] observation statistics (if objfn) | time-slice hand-over | slice flags |
// code words of the pair blocks (fast summary / raw runs over whole intervals of a multiple of eight steps)
static size_t header_bytes(int64_t n_catchments)
{
    return ((size_t)(kHdrInts + n_catchments) * sizeof(int) + 255) / 256 * 256;
}

static size_t obs_stats_bytes(const SmartEnsemble *e)
{
    if (!e->objfn)
        return 0;
    const int64_t R = smart_n_reports(e->n_steps, e->report_gap, e->report_type);
    return (size_t)e->n_catchments * (size_t)(kWsHead + R) * sizeof(double);
}

static size_t slice_bytes(int64_t n_samples, int64_t n_catchments)
{
    const size_t blocks = (size_t)((n_samples + kWave - 1) / kWave) * (size_t)n_catchments;
    return blocks * kSegFields * kWave * sizeof(double) + (blocks + 1) / 2 * 2 * sizeof(int);
}

static int merged_report(const SmartEnsemble *e);
static int plan_time_slices(const SmartEnsemble *e, int n_simd, int *per_simd, double *load);

// The streaming step loops jump through byte offsets that another kernel wrote (smart_fast_arms.h: pair blocks): right for
// a library whose code smartpy_amd.isa_lint has looked at.  smartpy_amd.build lints every library it links and, when the
// pair blocks lie where the code words point, writes "pairs-ok" over the "unchecked" of this stamp IN THE FILE (round 6:
// round 5 kept the verdict in a record next to the library, which a caller of the C ABI never reads -- advisor).  A
// library that carries no such stamp -- built by another route, by another hipcc -- runs the THREADED CHUNKS (the same
// arithmetic, bit for bit, no computed jumps) unless the environment says SMART_PAIR_BLOCKS=1; SMART_PAIR_BLOCKS=0 asks
// for the threaded chunks whatever the stamp (A/B runs of the tools).
extern "C" __attribute__((used, visibility("default"))) volatile char smart_lint_stamp[40] = "SMART_LINT_STAMP=unchecked";

static bool pair_blocks_wanted()
{
    if (const char *env = getenv("SMART_PAIR_BLOCKS"))
        return atoi(env) != 0;
    static const char ok[] = "SMART_LINT_STAMP=pairs-ok";
    for (size_t i = 0; i + 1 < sizeof(ok); ++i)
        if (smart_lint_stamp[i] != ok[i])
            return false;
    return true;
}

// what a sliced launch of this call needs for its hand-over (0: the call is not sliced, or no device to ask)
static size_t slices_need(const SmartEnsemble *e)
{
    if (e->math_mode != SMART_MATH_FAST)
        return 0;
    int per_simd = 0, n_dev = 0;
    double load = 0.0;
    DeviceCtx *d = hipGetDeviceCount(&n_dev) == hipSuccess && n_dev > 0 ? device_ctx() : nullptr;
    return d && plan_time_slices(e, d->n_simd, &per_simd, &load) > 1 ? slice_bytes(e->n_samples, e->n_catchments) : 0;
}

// the stream of records (rain, PE, observation, deviation per step) instead of the pair blocks' code words: a report every
// step, and the summary / raw reports whose gap is not a whole number of chunks
static bool uses_records(const SmartEnsemble *e)
{
    const int merged = merged_report(e);
    return merged == kReportEvery || ((merged == kReportMean || merged == kReportLast) && e->report_gap % kChunk != 0);
}

// the code words of the pair blocks: for the calls whose regular rows may take the streaming step loop; for a report every
// step (gap 1, where there are no pair blocks of that kind) the stream of records and its code words instead
static size_t codes_bytes(const SmartEnsemble *e)
{
    if (e->math_mode == SMART_MATH_FAST && uses_records(e))
        return ((size_t)e->n_catchments * (size_t)every_pairs(e->n_steps) * (8 * sizeof(double) + sizeof(unsigned)) + 255) /
               256 * 256;
    if (e->math_mode != SMART_MATH_FAST || e->report_gap % kChunk != 0)
        return 0;
    const int merged = merged_report(e);
    if (merged != kReportMean && merged != kReportLast)
        return 0;
    return ((size_t)e->n_catchments * (size_t)code_chunks(e->n_steps) * sizeof(uint2) + 255) / 256 * 256;
}

static int check(const SmartEnsemble *e)
{
    if (!e)
        return fail(SMART_E_NULL, "SmartEnsemble pointer is NULL");
    if (e->n_catchments < 1 || e->n_samples < 1 || e->n_steps < 1 || e->n_warm < 0 || e->report_gap < 1)
        return fail(SMART_E_SIZE, "sizes must satisfy n_catchments, n_samples, n_steps, report_gap >= 1 and n_warm >= 0 "
                                  "(got %lld, %lld, %lld, %lld, %lld)",
                    (long long)e->n_catchments, (long long)e->n_samples, (long long)e->n_steps,
                    (long long)e->report_gap, (long long)e->n_warm);
    if (e->report_type != SMART_REPORT_SUMMARY && e->report_type != SMART_REPORT_RAW)
        return fail(SMART_E_REPORT_TYPE, "Reporting type '%d' unknown.", (int)e->report_type);
    if (e->math_mode != SMART_MATH_LITERAL && e->math_mode != SMART_MATH_FAST)
        return fail(SMART_E_MODE, "math mode '%d' unknown.", (int)e->math_mode);
    if (e->n_warm > e->n_steps)
        return fail(SMART_E_WARMUP, "The warm-up duration (%lld steps) cannot exceed the length of the simulation period "
                                    "(%lld steps) because the beginning of the simulation period is used as made-up "
                                    "warm-up data.",
                    (long long)e->n_warm, (long long)e->n_steps);
    if (e->report_type == SMART_REPORT_SUMMARY && (e->n_steps % e->report_gap || e->n_warm % e->report_gap))
        return fail(SMART_E_GAP, "summary report: simulation length %lld and warm-up length %lld must be multiples of the "
                                 "report gap %lld",
                    (long long)e->n_steps, (long long)e->n_warm, (long long)e->report_gap);
    if (!(e->delta_sec > 0.0))
        return fail(SMART_E_SIZE, "delta_sec must be > 0");
    if (!e->area_m2 || !e->forcing || !e->params || !e->gw)
        return fail(SMART_E_NULL, "area_m2, forcing, params and gw are required");
    if (e->params_catchment_stride != 0 && e->params_catchment_stride < e->n_samples * 10)
        return fail(SMART_E_SIZE, "params_catchment_stride must be 0 or >= n_samples * 10");
    if (e->discharge && e->discharge_ld < e->n_samples)
        return fail(SMART_E_SIZE, "discharge_ld must be >= n_samples");
    if (e->objfn && (!e->obs || !e->workspace))
        return fail(SMART_E_NULL, "objfn needs obs and workspace");
    if (e->objfn && e->workspace_bytes < (int64_t)(header_bytes(e->n_catchments) + obs_stats_bytes(e)))
        return fail(SMART_E_SIZE, "workspace_bytes %lld is less than the %lld the header and the observation statistics need",
                    (long long)e->workspace_bytes, (long long)(header_bytes(e->n_catchments) + obs_stats_bytes(e)));
    if (e->time_slices < 0)
        return fail(SMART_E_SIZE, "time_slices must be >= 0");
    if (e->plan != 0 && !(e->plan & SMART_PLAN_VALID))
        return fail(SMART_E_SIZE, "plan must be 0 or a value returned by smart_plan_ensemble");
    if (e->literal_form < SMART_LITERAL_FORM_AUTO || e->literal_form > SMART_LITERAL_FORM_LANES)
        return fail(SMART_E_MODE, "literal_form must be SMART_LITERAL_FORM_AUTO, _ROWS or _LANES");
    g_err[0] = 0;
    return SMART_OK;
}

static int device_ready()
{
    int n = 0;
    hipError_t err = hipGetDeviceCount(&n);
    if (err != hipSuccess)
        return hip_fail(err, "hipGetDeviceCount");
    if (n < 1)
        return fail(SMART_E_NO_DEVICE, "no HIP device is visible; this engine has no CPU fallback");
    return SMART_OK;
}

// Which of the merged (sliceable) families take the regular rows of a call (run_ensemble_merged<..., REPORT>):
//   kReportMean   summary reports over intervals of two or more steps
//   kReportLast   raw reports over whole intervals (W and T multiples of the gap), final state vector not asked for
//   kReportEvery  a report every step (gap 1, either type), final state vector not asked for
// -1: none (smart_fast_plain: raw reports over a ragged time axis; raw / every-step reports with the final state vector)
static int merged_report(const SmartEnsemble *e)
{
    if (e->report_type == SMART_REPORT_SUMMARY && e->report_gap >= 2)
        return kReportMean;
    if (e->final_vars)
        return -1;
    if (e->report_gap == 1)
        return kReportEvery;
    if (e->report_type == SMART_REPORT_RAW && e->n_steps % e->report_gap == 0 && e->n_warm % e->report_gap == 0)
        return kReportLast;
    return -1;
}

// Time-sliced launch or not?  Whenever there are more blocks of 64 samples than SIMDs and the interval engine's
// preconditions hold (summary report, merged variant): pinned to one SIMD for the whole run, B blocks on S SIMDs last
// ceil(B / S) block-times -- 1e5 samples: 1,563 on 1,024 -> 2 against the 1.53 of a perfect split -- and above the
// residency limit the launch ends with a tail of whole blocks; sliced 16 ways the hardware dispatcher evens both out
// (measured -14 % at 1e5 samples, -33 % at 1.4e5, -17 % at 4e5; tools/debug/time_slices_sweep.py).  At or below one
// block per SIMD there is nothing to even out and the hand-over costs 10 %.  e->time_slices (or, for the tuning
// scripts under tools/, SMART_TIME_SLICES when that field is 0) overrides.
static int plan_time_slices(const SmartEnsemble *e, int n_simd, int *per_simd, double *load)
{
    const long blocks = (e->n_samples + kWave - 1) / kWave * e->n_catchments;
    const long cap = (blocks + n_simd - 1) / n_simd;
    *per_simd = (int)(cap < 1 ? 1 : cap); // blocks of 64 samples per SIMD, rounded up
    *load = (double)blocks / (double)n_simd;
    if (merged_report(e) < 0)
        return 1;
    const long n_all = e->n_warm / e->report_gap + e->n_steps / e->report_gap;
    int forced = e->time_slices;
    if (forced == 0) {
        const char *env = getenv("SMART_TIME_SLICES");
        if (env)
            forced = atoi(env) <= 0 ? 1 : atoi(env);
    }
    if (forced == 1 || n_all < 64)
        return 1;
    if (forced > 1)
        return forced < n_all / 4 ? forced : (int)(n_all / 4);
    // nothing to even out at or below one block per SIMD; and beyond ~48 per SIMD (3e6 samples) the ragged end of an
    // unsliced launch is under 2 % of it while the hand-over buffer (11 KB per block) starts to count
    if (blocks <= n_simd || blocks > 48L * n_simd)
        return 1;
    // with ten or more blocks per SIMD eight slices even the launch out as well as sixteen (tools/ab_slices.sh: 1e6
    // samples, 64 x 1e4) and halve the hand-over traffic, which is all the HBM traffic of a launch that stores no
    // discharge matrix (6.2 GB -> 3 GB per launch at 1e6 samples)
    // (below that, 24: level with 16 on the interval engine, 1 % ahead on the run engine and on the step loop since
    // its pair blocks made a slice cheaper -- tools/gpu_round.sh slices-flat, gpu_round.sh slices-headline)
    const long want = blocks >= 10L * n_simd ? 8 : 24;
    return (int)(n_all / 64 < want ? n_all / 64 : want);
}

// the pieces of e->workspace
struct Workspace {
    int *hdr = nullptr, *fflags = nullptr;
    double *stats = nullptr;
    char *slices = nullptr;
    size_t slice_room = 0;
    uint2 *codes = nullptr; // behind the hand-over, when the workspace has the room smart_workspace_bytes() asks for
};

static Workspace carve(const SmartEnsemble *e)
{
    Workspace w;
    const size_t hb = header_bytes(e->n_catchments), sb = obs_stats_bytes(e);
    if (!e->workspace || e->workspace_bytes < (int64_t)hb)
        return w;
    char *base = (char *)e->workspace;
    w.hdr = (int *)base;
    w.fflags = w.hdr + kHdrInts;
    if ((size_t)e->workspace_bytes >= hb + sb) {
        w.stats = sb ? (double *)(base + hb) : nullptr;
        w.slices = base + hb + sb;
        w.slice_room = (size_t)e->workspace_bytes - hb - sb;
        const size_t cb = codes_bytes(e), sl = slices_need(e);
        if (cb && w.slice_room >= sl + cb) { // (behind the hand-over of a sliced launch; 8-byte aligned like it)
            w.codes = (uint2 *)(w.slices + sl);
            w.slice_room = sl;
        }
    }
    return w;
}

static KArgs kernel_args(const SmartEnsemble *e, const Workspace &w)
{
    KArgs a;
    a.N = e->n_samples;
    a.T = e->n_steps;
    a.W = e->n_warm;
    a.gap = e->report_gap;
    a.R = smart_n_reports(e->n_steps, e->report_gap, e->report_type);
    a.first_len = e->report_type == SMART_REPORT_RAW && e->n_steps % e->report_gap ? e->n_steps % e->report_gap
                                                                                 : e->report_gap;
    a.report_type = e->report_type;
    a.dt = e->delta_sec;
    a.area = e->area_m2;
    a.forcing = e->forcing;
    a.params = e->params;
    a.pstride_c = e->params_catchment_stride;
    a.extra = e->extra;
    a.initial = e->initial;
    a.obs = e->obs;
    a.gw_obs = e->gw_obs;
    a.ws = e->objfn ? w.stats : nullptr;
    a.discharge = e->discharge;
    a.ld = e->discharge_ld;
    a.gw = e->gw;
    a.objfn = e->objfn;
    a.final_vars = e->final_vars;
    a.np_mean = e->math_mode == SMART_MATH_LITERAL && e->report_type == SMART_REPORT_SUMMARY && a.gap >= 8 && a.gap <= 128;
    a.pair_stride = e->final_vars ? (int)kPairStrideSplit : (int)kPairStride;
    a.n_catch = e->n_catchments;
    a.n_blocks = (a.N + kWave - 1) / kWave;
    a.seg_blocks = a.n_blocks * a.n_catch;
    a.hdr = w.hdr;
    a.fflags = nullptr;
    // the run lengths the forcing is tested for: the kMaxDiv largest divisors of the report gap, largest first, down
    // to 2 -- gap / q for q = 1, 2, 3, ... (at most sqrt(gap) trial divisions, whatever the gap)
    const int merged = merged_report(e);
    if ((merged == kReportMean || merged == kReportLast) && a.gap <= 0x7fffffff) {
        for (long q = 1; q * q <= a.gap && a.n_div < kMaxDiv; ++q)
            if (a.gap % q == 0 && a.gap / q >= 2)
                a.div[a.n_div++] = (int)(a.gap / q);
        // ... and, if there is room left, the small ones q itself (below sqrt(gap)), largest first
        for (long q = (long)std::sqrt((double)a.gap) + 1; q >= 2 && a.n_div < kMaxDiv; --q)
            if (q * q < a.gap && a.gap % q == 0 && q < a.div[a.n_div - 1])
                a.div[a.n_div++] = (int)q;
    }
    return a;
}

static void reset_workspace(const Workspace &w, long n_catch, int *flags, long n_flags, hipStream_t s)
{
    const long n_hdr = kHdrInts + n_catch, n = n_hdr > n_flags ? n_hdr : n_flags;
    hipLaunchKernelGGL(smart_workspace_reset, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w.hdr, n_hdr, flags,
                       n_flags);
}

// which of the two uses of Workspace::codes this call makes (codes_bytes)
static void set_codes(const SmartEnsemble *e, KArgs *a, const Workspace &w)
{
    a->codes = nullptr, a->estream = nullptr, a->ecodes = nullptr;
    if (!w.codes || !pair_blocks_wanted())
        return;
    if (uses_records(e)) {
        a->estream = (const double *)w.codes;
        a->ecodes = (const unsigned *)(a->estream + (size_t)e->n_catchments * (size_t)every_pairs(e->n_steps) * 8);
    } else {
        a->codes = w.codes;
    }
}

static void scan_forcing(const SmartEnsemble *e, KArgs a, const Workspace &w, hipStream_t s)
{
    a.fflags = w.fflags;
    set_codes(e, &a, w);
    hipLaunchKernelGGL(smart_forcing_scan, dim3(64, (unsigned)e->n_catchments), dim3(256), 0, s, a,
                       reinterpret_cast<const double2 *>(e->forcing));
}

struct Launch {
    FastKernel k;
    bool sliced;
};

static hipError_t launch_kernel(FastKernel k, KArgs a, dim3 grid, size_t lds, hipStream_t s)
{
    const double2 *forcing = reinterpret_cast<const double2 *>(a.forcing);
    const double *obs = a.obs, *ws = a.ws;
    void *args[] = {&a, &forcing, &obs, &ws};
    return hipLaunchKernel(fast_kernel(k), grid, dim3(kWave), args, lds, s);
}

// What a SMART_MATH_FAST call launches: the kernels (smart_fast_entry.h), sliced or not, and the load figures behind
// the choices.  Shared by the launch itself and by smart_describe_launch.
struct Decision {
    Launch todo[kMaxTodo];
    int n_todo = 0;
    bool overflow = false;
    void push(FastKernel k, bool sliced)
    {
        if (n_todo < kMaxTodo)
            todo[n_todo++] = {k, sliced};
        else
            overflow = true;
    }
    int n_seg = 1, per_simd = 0, exits = 0;
    int class_mask = 0, pc_mask = 0;
    int report = -1;        // merged_report(): which merged family takes the regular rows (-1: smart_fast_plain)
    bool intervals = false; // report == kReportMean
    double load = 0.0; // blocks of 64 samples per SIMD
    long illcond_blocks = 0;  // class-3 blocks the launch expects (the plan's count; without one every block of the call)
    bool illcond_rows = true; // their kernel: one sample per DPP row (smart_fast_illcond) or per lane (.._lanes)
};

// ---- the form of the literal step inside the fast mode (class 3), chosen per launch (round 6) -------------------------
// smart_fast_illcond advances FOUR samples per wavefront in ~150 instructions a step, smart_fast_illcond_lanes SIXTY-FOUR
// in ~440: per sample-step the row form costs seven times the SIMD cycles, and wins only while its sixteen wavefronts
// per block of 64 samples find SIMDs of their own -- a wavefront alone on its SIMD is as fast as its instruction count
// (DESIGN.md section 4.7).  Measured (profiles/r06_daily_form.txt; daily ensembles of the default LHS ranges, 11.6 % of
// the rows in class 3, B blocks of them): class 3 alone, rows against lanes -- B = 55: 1.78 / 3.96 ms, 91: 3.54 / 4.00,
// 127: 3.52 / 3.98 (two rounds of row-form wavefronts over 1,024 SIMDs still beat one round of the lane form), 181:
// 5.26 / 4.12, 362: 10.4 / 4.14, 1,810: 49.3 / 6.4.  Beside the other classes' kernels of the same call the SIMDs are
// shared: whole ensemble, rows against lanes -- 3e4 samples (880 row-form wavefronts + 416 others) 2.27 / 3.88 ms, 5e4
// (1,456 + 692) 3.94 / 3.96, 7e4 (2,032 + 968) 5.29 / 4.66, 1e5 6.60 / 5.74, 1e6 57.6 / 11.4.  Hence: rows while ALL
// the wavefronts of the call -- sixteen per class-3 block, one per other block -- fit into kIllCondRowRounds rounds over
// the chip's SIMDs.  SMART_ILLCOND_FORM=rows|lanes and SmartEnsemble.literal_form override (tests, A/B).
constexpr int kIllCondRowRounds = 2;

static long count_illcond_blocks(const SmartEnsemble *e, int plan)
{
    const long counted = (plan >> SMART_PLAN_ILLCOND_BLOCKS_SHIFT) & SMART_PLAN_ILLCOND_BLOCKS_MAX;
    const long all = (long)((e->n_samples + kWave - 1) / kWave * e->n_catchments);
    // (no count: a plan made by hand -- smart_row_class -- or none at all; a saturated count: every block may be one)
    return counted > 0 && counted < SMART_PLAN_ILLCOND_BLOCKS_MAX ? counted : all;
}

static bool illcond_form(const SmartEnsemble *e, int n_simd, long blocks)
{
    if (e->literal_form == SMART_LITERAL_FORM_ROWS)
        return true;
    if (e->literal_form == SMART_LITERAL_FORM_LANES)
        return false;
    if (const char *env = getenv("SMART_ILLCOND_FORM")) {
        if (!strcmp(env, "rows"))
            return true;
        if (!strcmp(env, "lanes"))
            return false;
    }
    const long all = (long)((e->n_samples + kWave - 1) / kWave * e->n_catchments);
    return blocks * kIllCondWaves + (all - blocks) <= (long)kIllCondRowRounds * n_simd;
}

static int decide(const SmartEnsemble *e, const DeviceCtx *d, const Workspace &w, Decision *out)
{
    Decision &x = *out;
    x.report = merged_report(e);
    x.intervals = x.report == kReportMean; // the merged summary kernels apply
    const int plan = (e->plan & SMART_PLAN_VALID) ? e->plan : (0x3f | SMART_PLAN_FORCING_RUNS);
    x.n_seg = plan_time_slices(e, d->n_simd, &x.per_simd, &x.load);
    // the hand-over buffer of a time-sliced launch sits behind the observation statistics in the caller's workspace;
    // a workspace without room for it means a plain launch
    if (x.n_seg > 1 && (!w.slices || w.slice_room < slice_bytes(e->n_samples, e->n_catchments)))
        x.n_seg = 1;
    // Early exits inside every step of the interval engine (FastModel::kExits) cost a wavefront a taken branch where
    // they trigger; the straight-line kernels run their wet intervals in two modes instead (SMART_WET_MODES: one
    // taken branch per interval) and keep the asm loop.  The exits win once a SIMD holds three waves whose scalar
    // work and branches hide behind each other's vector work -- by 1.5 to 2.5 % from 3 blocks per SIMD on (3.05,
    // 4.6, 15: config 4 on one GPU, config 5), level at 2.44, and lose 6 % at 1.9 and 1.53; rows ordered or not
    // (tools/gpu_r03_l.sh, gpu_r03_n.sh: profiles/r03_ab_exits_modes.txt).
    const char *env = getenv("SMART_EXITS");
    x.exits = env ? atoi(env) != 0 : x.load > 2.5;
    x.class_mask = plan & 0xf;
    if (plan & SMART_PLAN_CLASS_REGULAR) {
        if (x.intervals) {
            if (plan & SMART_PLAN_FORCING_PIECEWISE) {
                x.push(e->final_vars ? kIntervalsStates : (x.exits ? kIntervalsExits : kIntervals), true);
                x.pc_mask |= 1;
            }
            if (plan & SMART_PLAN_FORCING_RUNS) {
                x.push(e->final_vars ? kRunsStates : (x.exits ? kRunsExits : kRuns), true);
                x.pc_mask |= 4;
            }
            if (plan & SMART_PLAN_FORCING_VARYING) {
                x.push(e->final_vars ? kStepsStates : kSteps, true);
                x.pc_mask |= 2;
            }
        } else if (x.report == kReportLast) {
            if (plan & SMART_PLAN_FORCING_PIECEWISE) {
                x.push(kIntervalsRaw, true);
                x.pc_mask |= 1;
            }
            if (plan & (SMART_PLAN_FORCING_RUNS | SMART_PLAN_FORCING_VARYING)) {
                x.push(kStepsRaw, true);
                x.pc_mask |= 6;
            }
        } else if (x.report == kReportEvery) {
            x.push(kStepsEvery, true);
            x.pc_mask = 7;
        } else {
            x.push(kPlain, false);
        }
    }
    if (plan & SMART_PLAN_CLASS_STIFF)
        x.push(kStiff, false);
    if (plan & SMART_PLAN_CLASS_GUARD)
        x.push(kGuard, false);
    if (plan & SMART_PLAN_CLASS_ILLCOND) {
        x.illcond_blocks = count_illcond_blocks(e, plan);
        x.illcond_rows = illcond_form(e, d->n_simd, x.illcond_blocks);
        x.push(x.illcond_rows ? kIllCond : kIllCondLanes, false);
    }
    if (x.overflow)
        return fail(SMART_E_SIZE, "internal: a call wants more than %d kernels", kMaxTodo);
    if (x.n_todo == 0)
        return fail(SMART_E_SIZE, "the plan names no class of rows");
    return SMART_OK;
}

static int run(const SmartEnsemble *e, bool literal_rows = false)
{
    int rc = check(e);
    if (rc)
        return rc;
    if ((rc = device_ready()))
        return rc;
    hipStream_t s = (hipStream_t)e->stream;
    const Workspace w = carve(e);
    KArgs a = kernel_args(e, w);

    if (e->objfn)
        hipLaunchKernelGGL(smart_obs_prepare, dim3((unsigned)e->n_catchments), dim3(256), 0, s, e->obs, a.R, w.stats);

    const dim3 grid((unsigned)a.n_blocks, (unsigned)e->n_catchments);
    if (e->math_mode == SMART_MATH_LITERAL) {
        if (w.hdr) // a clean status word for smart_launch_status (a fast launch before this one may have left its own)
            reset_workspace(w, e->n_catchments, nullptr, 0, s);
        a.hdr = nullptr;
        launch_literal(a, grid, a.np_mean ? (size_t)a.gap * kWave * sizeof(double) : 0, s, literal_rows);
        HIP_TRY(hipGetLastError());
        return SMART_OK;
    }

    DeviceCtx *d = device_ctx();
    if (!d)
        return fail(SMART_E_NO_DEVICE, "cannot query the current HIP device");
    Decision x;
    if ((rc = decide(e, d, w, &x)))
        return rc;
    const int n_seg = x.n_seg, n_todo = x.n_todo;
    a.exits = x.exits;
    a.class_mask = x.class_mask;
    a.pc_mask = x.pc_mask;

    // ---- workspace header, forcing flags, slice counters
    KArgs a_sliced = a;
    if (n_seg > 1) {
        a_sliced.n_seg = n_seg;
        a_sliced.seg_state = (double *)w.slices;
        a_sliced.seg_flag = (int *)(w.slices + (size_t)a.seg_blocks * kSegFields * kWave * sizeof(double));
        const char *mp = getenv("SMART_DEBUG_MAX_POLLS"), *dd = getenv("SMART_DEBUG_DROP_SLICE"); // tests only
        a_sliced.max_polls = mp && atol(mp) > 0 ? atol(mp) : kDefaultMaxPolls;
        a_sliced.debug_drop = dd ? atoi(dd) : 0;
    }
    if (w.hdr) {
        reset_workspace(w, e->n_catchments, a_sliced.seg_flag, n_seg > 1 ? a.seg_blocks : 0, s);
        // every fast launch looks at its forcing (the merged kernels for what the flags say about runs of equal
        // values; all of them for the NaN that belongs to the literal kernel: status word)
        scan_forcing(e, a, w, s);
        if (x.report >= 0 && (x.class_mask & SMART_PLAN_CLASS_REGULAR)) {
            a.fflags = a_sliced.fflags = w.fflags;
            set_codes(e, &a, w);
            a_sliced.codes = a.codes, a_sliced.estream = a.estream, a_sliced.ecodes = a.ecodes;
        }
    }

    // ---- launch: one kernel on the caller's stream; several fork onto the device's auxiliary streams and join
    std::lock_guard<std::mutex> lock(d->mu);
    auto launch = [&](const Launch &l, hipStream_t st) -> hipError_t {
        if (l.sliced && n_seg > 1) {
            // dynamic LDS is requested only to cap the resident workgroups at `per_simd` per SIMD (working + waiting;
            // beyond 3 the register file is the limit anyway)
            const size_t lds = x.per_simd <= 3 ? lds_for_residency(d, l.k, 4 * x.per_simd) : 0;
            return launch_kernel(l.k, a_sliced, dim3((unsigned)(a.seg_blocks * n_seg), 1), lds, st);
        }
        // (the ill-conditioned rows: one sample per DPP row, sixteen workgroups per block of 64 samples)
        return launch_kernel(l.k, a, l.k == kIllCond ? dim3(grid.x * kIllCondWaves, grid.y) : grid, 0, st);
    };
    if (n_todo == 1) {
        HIP_TRY(launch(x.todo[0], s));
        return SMART_OK;
    }
    if (!d->fork) {
        HIP_TRY(hipEventCreateWithFlags(&d->fork, hipEventDisableTiming));
        for (int i = 0; i < kMaxAux; ++i) {
            HIP_TRY(hipStreamCreateWithFlags(&d->aux[i], hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&d->join[i], hipEventDisableTiming));
        }
    }
    HIP_TRY(hipEventRecord(d->fork, s));
    for (int i = 1; i < n_todo; ++i) {
        HIP_TRY(hipStreamWaitEvent(d->aux[i - 1], d->fork, 0));
        HIP_TRY(launch(x.todo[i], d->aux[i - 1]));
        HIP_TRY(hipEventRecord(d->join[i - 1], d->aux[i - 1]));
    }
    HIP_TRY(launch(x.todo[0], s));
    for (int i = 1; i < n_todo; ++i)
        HIP_TRY(hipStreamWaitEvent(s, d->join[i - 1], 0));
    return SMART_OK;
}

static int describe(const SmartEnsemble *e, char *text, int64_t len)
{
    if (!text || len < 1)
        return fail(SMART_E_NULL, "smart_describe_launch: no room for the text");
    text[0] = 0;
    int rc = check(e);
    if (rc)
        return rc;
    if (e->math_mode == SMART_MATH_LITERAL) {
        snprintf(text, (size_t)len, "smart_ensemble_literal");
        return SMART_OK;
    }
    if ((rc = device_ready()))
        return rc;
    DeviceCtx *d = device_ctx();
    if (!d)
        return fail(SMART_E_NO_DEVICE, "cannot query the current HIP device");
    Decision x;
    if ((rc = decide(e, d, carve(e), &x)))
        return rc;
    const long blocks = (e->n_samples + kWave - 1) / kWave * e->n_catchments;
    size_t used = 0;
    for (int i = 0; i < x.n_todo && used + 1 < (size_t)len; ++i) {
        int n;
        if (x.todo[i].sliced && x.n_seg > 1)
            n = snprintf(text + used, (size_t)len - used, "%s%s[%d slices x %ld blocks, %d resident per SIMD]",
                         i ? " + " : "", kFastKernelNames[x.todo[i].k], x.n_seg, blocks, x.per_simd);
        else if (x.todo[i].k == kIllCond || x.todo[i].k == kIllCondLanes)
            // (the class-3 blocks the form was chosen for: the plan's count, or every block where there is none)
            n = snprintf(text + used, (size_t)len - used, "%s%s[%ld of %ld blocks, one sample per %s]", i ? " + " : "",
                         kFastKernelNames[x.todo[i].k], x.illcond_blocks, blocks,
                         x.illcond_rows ? "DPP row x 16 wavefronts" : "lane");
        else
            n = snprintf(text + used, (size_t)len - used, "%s%s[%ld blocks]", i ? " + " : "",
                         kFastKernelNames[x.todo[i].k], blocks);
        if (n < 0)
            break;
        used += (size_t)n;
    }
    return SMART_OK;
}

static int64_t workspace_bytes(const SmartEnsemble *e)
{
    if (!e || e->n_catchments < 1 || e->n_samples < 1 || e->n_steps < 0 || e->report_gap < 1)
        return 0;
    return (int64_t)(header_bytes(e->n_catchments) + obs_stats_bytes(e) + slices_need(e) + codes_bytes(e));
}

static int make_plan(const SmartEnsemble *e, int32_t *plan)
{
    if (!plan)
        return fail(SMART_E_NULL, "smart_plan_ensemble: plan is NULL");
    *plan = 0;
    int rc = check(e);
    if (rc)
        return rc;
    if ((rc = device_ready()))
        return rc;
    const Workspace w = carve(e);
    if (!w.hdr)
        return fail(SMART_E_NULL, "smart_plan_ensemble needs a workspace of smart_workspace_bytes() bytes");
    hipStream_t s = (hipStream_t)e->stream;
    KArgs a = kernel_args(e, w);
    reset_workspace(w, e->n_catchments, nullptr, 0, s);
    hipLaunchKernelGGL(smart_classify_rows, dim3((unsigned)a.n_blocks, (unsigned)e->n_catchments), dim3(kWave), 0, s, a);
    if (merged_report(e) == kReportMean || merged_report(e) == kReportLast) {
        scan_forcing(e, a, w, s);
        a.fflags = w.fflags;
        hipLaunchKernelGGL(smart_classify_forcing, dim3((unsigned)((e->n_catchments + 255) / 256)), dim3(256), 0, s,
                           a, w.hdr);
    }
    HIP_TRY(hipGetLastError());
    int bits[2] = {0, 0};
    static_assert(kHdrPlanIllCond == kHdrPlan + 1, "read back together");
    HIP_TRY(hipMemcpyAsync(bits, w.hdr + kHdrPlan, sizeof(bits), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    const long n_ill = bits[1] < 0 ? 0 : (bits[1] > SMART_PLAN_ILLCOND_BLOCKS_MAX ? SMART_PLAN_ILLCOND_BLOCKS_MAX : bits[1]);
    *plan = SMART_PLAN_VALID | (bits[0] & (0x3f | SMART_PLAN_FORCING_RUNS)) |
            (int32_t)(n_ill << SMART_PLAN_ILLCOND_BLOCKS_SHIFT);
    return SMART_OK;
}

static int launch_status(const SmartEnsemble *e, int32_t *status)
{
    if (!e || !status)
        return fail(SMART_E_NULL, "smart_launch_status: NULL argument");
    *status = 0;
    if (!e->workspace || e->workspace_bytes < (int64_t)header_bytes(e->n_catchments < 1 ? 1 : e->n_catchments))
        return SMART_OK;
    int rc = device_ready();
    if (rc)
        return rc;
    int word = 0;
    HIP_TRY(hipMemcpyAsync(&word, (const int *)e->workspace + kHdrStatus, sizeof(int), hipMemcpyDeviceToHost,
                           (hipStream_t)e->stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)e->stream));
    *status = word;
    return SMART_OK;
}

// ---- smartcpp.allsteps (structure.py:56-62,118-121,143-146) ---------------------------------------------------------
// The reference calls it twice per SMART.simulate() -- the warm-up over the first W steps of the series, then the run
// over all of it -- and a calibration loop calls simulate() thousands of times with the SAME series.  Round 3 paid a
// hipMalloc, an upload of the whole series and a hipFree per call.  Now the device buffers live as long as the
// library, and the series is uploaded once: the library keeps a host copy of what is on the device and compares
// (memcmp: ~50 us for ten years of hourly values) -- a call whose rain / peva start with what is cached uploads only
// what lies beyond (the run after its warm-up: the rest of the series; the next simulate(): nothing).
// One caller at a time (the reference holds the GIL across the call); g_hook_mu makes that a guarantee.
struct HookCache {
    int device = -1;
    double *forcing = nullptr; // [cap][2] on the device
    size_t cap = 0;            // doubles the device buffer holds (two per step)
    std::vector<double> rain, peva; // host copies of the steps that are on the device
    double *io = nullptr;      // params[10] initial[12] area[1] | discharge[R] gw[1] final[19] | header
    size_t io_cap = 0;         // doubles
    std::vector<double> stage;
    int64_t n_calls = 0, n_malloc = 0, bytes_up = 0, n_fast = 0;
    // SMART_ALLSTEPS_MATH=fast: what smart_plan_ensemble said about the first `len` steps of the cached series under a
    // report (gap, type) -- the kinds of forcing it holds, and whether the launch came back with a status word (a NaN or
    // an infinity in the series: the literal kernel's business).  A calibration loop calls with the same (len, gap,
    // type) thousands of times (two of them per simulate(): warm-up, run); the entries die with the cached series.
    struct Known {
        int64_t len, gap;
        int type, forcing_bits;
        bool literal_only;
    };
    std::vector<Known> known;
    int64_t n_plans = 0;
};
static HookCache g_hook;
static std::mutex g_hook_mu;
constexpr size_t kHookHeaderDoubles = 64; // 512 bytes: the workspace header of a one-catchment call (status word)

static int hook_reserve(double **buf, size_t *cap, size_t want)
{
    if (*cap >= want)
        return SMART_OK;
    if (*buf)
        (void)hipFree(*buf);
    *buf = nullptr;
    *cap = 0;
    const size_t room = want + want / 2 + 1024;
    HIP_TRY(hipMalloc(buf, room * sizeof(double)));
    ++g_hook.n_malloc;
    *cap = room;
    return SMART_OK;
}

// wave_class() (smart_fast_model.h) for ONE row, on the host: which arithmetic class its kernel is.  The device's own
// classification stays authoritative -- a row this function puts into the wrong class meets no kernel, raises
// SMART_STATUS_STALE_PLAN, and the call repeats in literal arithmetic -- so what is needed here is agreement on ordinary
// rows, which the rules being the same ones gives.
static int host_row_class(const double *p, double dt, const double *st12, double area)
{
    auto nonfinite = [](double x) {
        unsigned long long u;
        std::memcpy(&u, &x, sizeof(u));
        return (u & 0x7ff0000000000000ull) == 0x7ff0000000000000ull;
    };
    const bool stiff = !(p[6] * 3600.0 >= dt && p[7] * 3600.0 >= dt && p[8] * 3600.0 >= dt && p[9] * 3600.0 >= dt);
    const bool guard = !(p[4] >= 0.0 && p[4] <= 0.5 && p[1] >= 0.0 && p[5] > 0.0);
    bool wild = !(p[9] * 3600.0 >= 0.5 * dt);
    for (int i = 0; i < 10; ++i)
        wild = wild || nonfinite(p[i]);
    wild = wild || !(p[3] >= 0.0 && p[3] <= 1.0) || !(p[2] >= 0.0 && p[2] <= 1.0) || !(p[0] >= 0.0);
    wild = wild || !(p[6] > 0.0 && p[7] > 0.0 && p[8] > 0.0 && p[9] > 0.0);
    wild = wild || !(p[5] > 0.0) || !(p[0] >= 0.2) || !(p[5] <= 1.0e3) || !(p[5] >= 1.0);
    if (st12) {
        double lay = 0.0;
        for (int i = 0; i < 12; ++i) {
            unsigned long long u;
            std::memcpy(&u, &st12[i], sizeof(u));
            wild = wild || nonfinite(st12[i]) || u > 0x8000000000000000ull;
            if (i >= 5 && i < 11)
                lay += st12[i];
        }
        if (!wild) {
            const double fill = (lay / area * 1e3) / p[5];
            const double s_init = p[4] * fill, h_init = p[2] * fill;
            wild = !(s_init <= 0.5) || !(h_init <= 1.0); // (a NaN fails both compares)
        }
    }
    return wild ? 3 : (guard ? 2 : (stiff ? 1 : 0));
}

static int allsteps(double area_m2, double delta_sec, int64_t length_simu, const double *nd_rain,
                    const double *nd_peva, const double *nd_parameters, const double *nd_initial, int32_t report_type,
                    int64_t report_gap, double *discharge, double *groundwater_component, double *final_vars)
{
    if (!nd_rain || !nd_peva || !nd_parameters || !nd_initial || !discharge || !groundwater_component || !final_vars)
        return fail(SMART_E_NULL, "smart_allsteps_hip: NULL argument");
    if (length_simu < 1 || report_gap < 1)
        return fail(SMART_E_SIZE, "smart_allsteps_hip: length_simu and report_gap must be >= 1");
    int rc = device_ready();
    if (rc)
        return rc;
    std::lock_guard<std::mutex> lock(g_hook_mu);
    HookCache &h = g_hook;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (h.device != dev) { // (another device than last time: its buffers are of no use here)
        if (h.forcing)
            (void)hipFree(h.forcing);
        if (h.io)
            (void)hipFree(h.io);
        h.forcing = h.io = nullptr;
        h.cap = h.io_cap = 0;
        h.rain.clear();
        h.peva.clear();
        h.known.clear();
        h.device = dev;
    }
    ++h.n_calls;
    const size_t L = (size_t)length_simu;
    const int64_t R = smart_n_reports(length_simu, report_gap, report_type);
    const size_t n_rep = (size_t)(R > 0 ? R : 1);

    // ---- the series: what of it is on the device already?
    const size_t have = h.rain.size() < L ? h.rain.size() : L;
    if (have && (std::memcmp(h.rain.data(), nd_rain, have * sizeof(double)) ||
                 std::memcmp(h.peva.data(), nd_peva, have * sizeof(double)))) {
        h.rain.clear();
        h.peva.clear();
        h.known.clear();
    }
    if (2 * L > h.cap) { // a longer series than the buffer holds: a new buffer, everything uploaded again.  The first
                         // one has room for 15 years of hourly steps: the warm-up call that usually comes first is short
        const size_t floor_steps = 131072;
        if ((rc = hook_reserve(&h.forcing, &h.cap, 2 * (L > floor_steps ? L : floor_steps))))
            return rc;
        h.rain.clear();
        h.peva.clear();
        h.known.clear();
    }
    if (h.rain.size() < L) {
        const size_t from = h.rain.size(), n = L - from;
        h.stage.resize(2 * n);
        for (size_t t = 0; t < n; ++t) {
            h.stage[2 * t] = nd_rain[from + t];
            h.stage[2 * t + 1] = nd_peva[from + t];
        }
        HIP_TRY(hipMemcpy(h.forcing + 2 * from, h.stage.data(), 2 * n * sizeof(double), hipMemcpyHostToDevice));
        h.bytes_up += (int64_t)(2 * n * sizeof(double));
        h.rain.insert(h.rain.end(), nd_rain + from, nd_rain + L);
        h.peva.insert(h.peva.end(), nd_peva + from, nd_peva + L);
    }

    // ---- the small block: parameters, states, area in; discharge, ratio, final row out; a workspace header
    // (SMART_ALLSTEPS_MATH=fast: room for what the fast launch wants beside the header -- the code words of the step loop)
    const char *math = getenv("SMART_ALLSTEPS_MATH");
    const bool fast = math && std::strcmp(math, "fast") == 0;
    size_t ws_doubles = kHookHeaderDoubles;
    if (fast) {
        SmartEnsemble sized;
        std::memset(&sized, 0, sizeof(sized));
        sized.n_catchments = sized.n_samples = 1;
        sized.n_steps = length_simu;
        sized.report_gap = report_gap;
        sized.report_type = report_type;
        sized.math_mode = SMART_MATH_FAST;
        sized.time_slices = 1;
        sized.final_vars = (double *)1; // (asked for: decides which kernels, and with them what the workspace holds)
        const size_t want = ((size_t)workspace_bytes(&sized) + 7) / 8;
        ws_doubles = want > ws_doubles ? want : ws_doubles;
    }
    const size_t n_in = 10 + 12 + 1, n_out = n_rep + 1 + 19;
    if ((rc = hook_reserve(&h.io, &h.io_cap, n_in + n_out + ws_doubles)))
        return rc;
    double in[n_in];
    std::memcpy(in, nd_parameters, 10 * sizeof(double));
    std::memcpy(in + 10, nd_initial + 7, 12 * sizeof(double)); // only the states are read (structure.py:182-187)
    in[22] = area_m2;
    HIP_TRY(hipMemcpy(h.io, in, n_in * sizeof(double), hipMemcpyHostToDevice));

    SmartEnsemble e;
    std::memset(&e, 0, sizeof(e));
    e.n_catchments = 1;
    e.n_samples = 1;
    e.n_steps = length_simu;
    e.n_warm = 0;
    e.report_gap = report_gap;
    e.report_type = report_type;
    e.delta_sec = delta_sec;
    e.forcing = h.forcing;
    e.params = h.io;
    e.initial = h.io + 10;
    e.area_m2 = h.io + 22;
    double *o = h.io + n_in;
    e.discharge = o;
    e.discharge_ld = 1;
    e.gw = o + n_rep;
    e.final_vars = e.gw + 1;
    e.workspace = o + n_out;
    e.workspace_bytes = (int64_t)(ws_doubles * sizeof(double));
    e.time_slices = 1;
    // SMART_ALLSTEPS_MATH=fast: the fast kernels for this one sample (interval engine / step loop, SPLIT: the final row is
    // asked for) -- <= 1e-9 of the reference instead of its bits, at a tenth of the time.  Default: literal arithmetic.
    e.math_mode = fast ? SMART_MATH_FAST : SMART_MATH_LITERAL;
    if (fast) {
        // ONE kernel for ONE row: the row's class worked out here (ten numbers), the kinds of forcing of this series
        // remembered from the first call with this (length, gap, report type).  Round 4 left e.plan at 0: six kernels,
        // the forcing scan, an auxiliary-stream fork and join per call -- for one sample, thousands of times.
        HookCache::Known *k = nullptr;
        for (auto &x : h.known)
            if (x.len == length_simu && x.gap == report_gap && x.type == report_type)
                k = &x;
        if (!k) {
            int32_t plan = 0;
            if ((rc = make_plan(&e, &plan)))
                return rc;
            ++h.n_plans;
            if (h.known.size() >= 8)
                h.known.erase(h.known.begin());
            h.known.push_back({length_simu, report_gap, report_type,
                               plan & (SMART_PLAN_FORCING_PIECEWISE | SMART_PLAN_FORCING_VARYING | SMART_PLAN_FORCING_RUNS),
                               false});
            k = &h.known.back();
        }
        if (k->literal_only) { // (the series is known to hold a NaN or an infinity: straight to the literal kernel)
            e.math_mode = SMART_MATH_LITERAL;
        } else {
            const int cls = host_row_class(nd_parameters, delta_sec, nd_initial + 7, area_m2);
            e.plan = SMART_PLAN_VALID | k->forcing_bits | (1 << cls);
        }
    }
    rc = run(&e, /*rows=*/true);
    if (rc == SMART_OK && e.math_mode == SMART_MATH_FAST) {
        ++h.n_fast;
        int32_t word = 0;
        rc = launch_status(&e, &word);
        if (rc == SMART_OK && word != 0) { // (a NaN in the series, a row this side put into the wrong class: literal)
            if (word & SMART_STATUS_NONFINITE_FORCING)
                for (auto &x : h.known)
                    if (x.len == length_simu && x.gap == report_gap && x.type == report_type)
                        x.literal_only = true;
            e.math_mode = SMART_MATH_LITERAL;
            rc = run(&e, true);
        }
    }
    if (rc != SMART_OK)
        return rc;
    h.stage.resize(n_out);
    HIP_TRY(hipMemcpy(h.stage.data(), o, n_out * sizeof(double), hipMemcpyDeviceToHost));
    std::memcpy(discharge, h.stage.data(), (size_t)(R > 0 ? R : 0) * sizeof(double));
    *groundwater_component = h.stage[n_rep];
    std::memcpy(final_vars, h.stage.data() + n_rep + 1, 19 * sizeof(double));
    g_err[0] = 0;
    return SMART_OK;
}

} // namespace smart

using namespace smart;

extern "C" {

int64_t smart_n_reports(int64_t n_steps, int64_t report_gap, int32_t report_type)
{
    if (report_gap < 1 || n_steps < 0)
        return 0;
    return report_type == SMART_REPORT_RAW ? (n_steps + report_gap - 1) / report_gap : n_steps / report_gap;
}

int smart_check_ensemble(const SmartEnsemble *e) { return check(e); }

int64_t smart_workspace_bytes(const SmartEnsemble *e) { return workspace_bytes(e); }

int smart_run_ensemble_hip(const SmartEnsemble *e) { return run(e); }

int smart_plan_ensemble(const SmartEnsemble *e, int32_t *plan) { return make_plan(e, plan); }

int smart_launch_status(const SmartEnsemble *e, int32_t *status) { return launch_status(e, status); }

int smart_describe_launch(const SmartEnsemble *e, char *text, int64_t len) { return describe(e, text, len); }

int smart_allsteps_hip(double area_m2, double delta_sec, int64_t length_simu, const double *nd_rain,
                       const double *nd_peva, const double *nd_parameters, const double *nd_initial,
                       int32_t report_type, int64_t report_gap, double *discharge, double *groundwater_component,
                       double *final_vars)
{
    return allsteps(area_m2, delta_sec, length_simu, nd_rain, nd_peva, nd_parameters, nd_initial, report_type,
                    report_gap, discharge, groundwater_component, final_vars);
}

int smart_hook_counters(int64_t *counters, int64_t n)
{
    if (!counters || n < 1)
        return fail(SMART_E_NULL, "smart_hook_counters: no room for the counters");
    std::lock_guard<std::mutex> lock(g_hook_mu);
    const int64_t v[5] = {g_hook.n_calls, g_hook.n_malloc, g_hook.bytes_up, g_hook.n_fast, g_hook.n_plans};
    for (int64_t i = 0; i < n && i < 5; ++i)
        counters[i] = v[i];
    return SMART_OK;
}

int smart_onestep_hip(int64_t n, const double *in, double *out)
{
    if (!in || !out)
        return fail(SMART_E_NULL, "smart_onestep_hip: NULL argument");
    if (n < 1)
        return fail(SMART_E_SIZE, "smart_onestep_hip: n must be >= 1");
    int rc = device_ready();
    if (rc)
        return rc;
    double *dev = nullptr;
    HIP_TRY(hipMalloc(&dev, (size_t)n * (26 + 19) * sizeof(double)));
    hipError_t err = hipMemcpy(dev, in, (size_t)n * 26 * sizeof(double), hipMemcpyHostToDevice);
    if (err == hipSuccess) {
        launch_onestep(n, dev, dev + n * 26, nullptr);
        err = hipGetLastError();
    }
    if (err == hipSuccess)
        err = hipMemcpy(out, dev + n * 26, (size_t)n * 19 * sizeof(double), hipMemcpyDeviceToHost);
    (void)hipFree(dev);
    if (err != hipSuccess)
        return hip_fail(err, "smart_onestep_hip");
    g_err[0] = 0;
    return SMART_OK;
}

int smart_river_step_hip(int64_t n, const double *in, double *out)
{
    if (!in || !out)
        return fail(SMART_E_NULL, "smart_river_step_hip: NULL argument");
    if (n < 1)
        return fail(SMART_E_SIZE, "smart_river_step_hip: n must be >= 1");
    int rc = device_ready();
    if (rc)
        return rc;
    double *dev = nullptr;
    HIP_TRY(hipMalloc(&dev, (size_t)n * (4 + 2) * sizeof(double)));
    hipError_t err = hipMemcpy(dev, in, (size_t)n * 4 * sizeof(double), hipMemcpyHostToDevice);
    if (err == hipSuccess) {
        launch_river(n, dev, dev + n * 4, nullptr);
        err = hipGetLastError();
    }
    if (err == hipSuccess)
        err = hipMemcpy(out, dev + n * 4, (size_t)n * 2 * sizeof(double), hipMemcpyDeviceToHost);
    (void)hipFree(dev);
    if (err != hipSuccess)
        return hip_fail(err, "smart_river_step_hip");
    g_err[0] = 0;
    return SMART_OK;
}

int smart_objfn_hip(int64_t n_samples, int64_t n_reports, const double *sim, int64_t ld, const double *obs,
                    const double *gw_sim, double gw_obs, double *objfn, void *stream)
{
    if (!sim || !obs || !objfn)
        return fail(SMART_E_NULL, "smart_objfn_hip: sim, obs and objfn are required");
    if (n_samples < 1 || n_reports < 1 || ld < n_samples)
        return fail(SMART_E_SIZE, "smart_objfn_hip: need n_samples, n_reports >= 1 and ld >= n_samples");
    int rc = device_ready();
    if (rc)
        return rc;
    if (n_samples >= 4 * 65536) // >= 4 wavefronts per SIMD even with one lane per sample
        hipLaunchKernelGGL((smart_objfn_matrix<4, 1, 8>), dim3((unsigned)((n_samples + 4 * kWave - 1) / (4 * kWave))),
                           dim3(4 * kWave), 0, (hipStream_t)stream, (long)n_samples, (long)n_reports, sim, (long)ld, obs,
                           gw_sim, gw_obs, objfn);
    else
        hipLaunchKernelGGL((smart_objfn_matrix<1, 8, 4>), dim3((unsigned)((n_samples + kWave - 1) / kWave)),
                           dim3(8 * kWave), 0, (hipStream_t)stream, (long)n_samples, (long)n_reports, sim, (long)ld, obs,
                           gw_sim, gw_obs, objfn);
    HIP_TRY(hipGetLastError());
    g_err[0] = 0;
    return SMART_OK;
}

int smart_row_class(const double *params, double delta_sec, const double *initial12, double area_m2)
{
    if (!params)
        return fail(SMART_E_NULL, "smart_row_class: params is NULL");
    return host_row_class(params, delta_sec, initial12, area_m2);
}

int smart_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

int smart_abi_version(void) { return SMART_AMD_ABI_VERSION; }

#define SMART_STR_(x) #x
#define SMART_STR(x) SMART_STR_(x)
const char *smart_build_info(void)
{
    static char text[256];
    static std::once_flag once;
    std::call_once(once, [] {
        char stamp[40];
        for (size_t i = 0; i < sizeof(stamp); ++i)
            stamp[i] = smart_lint_stamp[i];
        stamp[sizeof(stamp) - 1] = 0;
        snprintf(text, sizeof(text), "hipcc %d.%d.%d | clang %s | gfx950 | ABI %d | %s", HIP_VERSION_MAJOR,
                 HIP_VERSION_MINOR, HIP_VERSION_PATCH, __clang_version__, SMART_AMD_ABI_VERSION, stamp);
    });
    return text;
}

const char *smart_last_error(void) { return g_err; }

} // extern "C"
