/*
 * smart_oracle.c -- CPU restatement of the SMART time loop.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle for the HIP ensemble engine in smartpy_amd/csrc.  It is a
 * plain-C, scalar, one-sample-at-a-time restatement of the reference algorithm
 * (ThibHlln/smartpy v0.2.2, smartpy/structure.py).  Nothing in the product path may include,
 * link, import or call it: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do,
 * and only as the checker / the reported CPU baseline.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function below against
 * vectors produced by importing the reference itself in the build container
 * (tests/golden/make_golden.py) and against the reference's own golden values
 * (tests/test_run_daily_to_hourly.py:30-122, examples/out/ExampleDaily/ExampleDaily.mod.flow).
 * With pow_mode = SMART_POW_LIBM and sum_mode = SMART_SUM_NUMPY the oracle is bit-identical to the
 * reference run in the same container (same glibc pow, numpy's pairwise reduction emulated).
 *
 * Citations "structure.py:NNN" are lines of /root/reference/smartpy/structure.py.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fno-builtin-pow -fopenmp).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define SMART_POW_LIBM 0 /* s' ** i through libm pow(), as CPython's float.__pow__ does             */
#define SMART_POW_MUL 1  /* s' ** i as the left-to-right product chain the HIP "literal" kernel uses */

#define SMART_SUM_NUMPY 0 /* report means / gw sums in numpy's pairwise order (bit-parity with ref)  */
#define SMART_SUM_SEQ 1   /* plain left-to-right sums (what a streaming GPU kernel can do)           */
#define SMART_SUM_GPU 2   /* what the HIP literal kernel does: numpy-ordered report means for gaps   */
                          /* up to 128 (it stages a report interval in LDS), sequential gw sums      */

#define SMART_REPORT_SUMMARY 1 /* structure.py:65-66 */
#define SMART_REPORT_RAW 2     /* structure.py:67-68 */

#define NVAR 19 /* 7 outputs + 12 states, order of structure.py:78-82 */

/* ---- s' ** i (structure.py:382, 396) ------------------------------------------------------- */
static double spow(double s, int i, int pow_mode)
{
    if (pow_mode == SMART_POW_LIBM)
        return pow(s, (double)i);
    double p = s;
    for (int k = 1; k < i; ++k)
        p = p * s;
    return p;
}

/*
 * One model step: catchment (structure.py:267-458) then river (structure.py:461-503), glued as
 * structure.py:200-264.  p = T,C,H,D,S,Z,SK,FK,GK,RK (parameters.py:25); st = V_ove,V_dra,V_int,
 * V_sgw,V_dgw,V_ly1..6,V_river; out = the 19-vector of structure.py:78-82.
 */
void smart_oracle_one_step(double area, double dt, double rain_in, double peva_in, const double *p,
                           const double *st, double *out, int pow_mode)
{
    const double pT = p[0], pC = p[1], pH = p[2], pD = p[3], pS = p[4], pZ = p[5];
    const double sk = p[6] * 3600.0; /* structure.py:320 */
    const double fk = p[7] * 3600.0; /* structure.py:321 */
    const double gk = p[8] * 3600.0; /* structure.py:322 */
    double v_ove = st[0], v_dra = st[1], v_int = st[2], v_sgw = st[3], v_dgw = st[4];

    double z[7], l[7];
    z[0] = 0.0;
    l[0] = 0.0;
    for (int i = 1; i <= 6; ++i) {
        z[i] = pZ / 6.0;                    /* structure.py:329-337 */
        l[i] = st[4 + i] / area * 1e3;      /* structure.py:339-347 */
    }
    double tot = 0.0; /* Python sum() starts from int 0: 0 + 0.0 = 0.0, then left to right (:350) */
    for (int i = 0; i <= 6; ++i)
        tot = tot + l[i];

    const double rain = rain_in * pT; /* structure.py:353 */
    double ex = rain - peva_in;       /* structure.py:355 */
    double aeva = 0.0;
    double of, df, inf, sh, dp;

    if (ex >= 0.0) { /* structure.py:359 */
        aeva += peva_in;
        const double hp = pH * (tot / pZ); /* :363 */
        of = hp * ex;                      /* :364 */
        ex -= of;                          /* :365 */
        for (int i = 1; i <= 6; ++i) {     /* :367-374 */
            const double sp = z[i] - l[i];
            if (ex <= sp) {
                l[i] += ex;
                ex = 0.0;
            } else {
                l[i] = z[i];
                ex -= sp;
            }
        }
        df = pD * ex;          /* :376 */
        inf = (1.0 - pD) * ex; /* :377 */
        const double s1 = pS * (tot / pZ); /* :379, tot from before infiltration */
        for (int i = 1; i <= 6; ++i) {     /* :381-385 */
            const double lk = l[i] * spow(s1, i, pow_mode);
            if (lk < l[i]) {
                inf += lk;
                l[i] -= lk;
            }
        }
        sh = 0.0;
        for (int i = 1; i <= 6; ++i) { /* :387-392 */
            const double lk = l[i] * (s1 / (double)i);
            if (lk < l[i]) {
                sh += lk;
                l[i] -= lk;
            }
        }
        dp = 0.0;
        for (int i = 6; i >= 1; --i) { /* :394-399 */
            const double lk = l[i] * spow(s1, 7 - i, pow_mode);
            if (lk < l[i]) {
                dp += lk;
                l[i] -= lk;
            }
        }
    } else { /* structure.py:400 */
        of = 0.0;
        df = 0.0;
        inf = 0.0;
        sh = 0.0;
        dp = 0.0;
        double d = ex * (-1.0); /* :407 */
        aeva += rain;
        for (int i = 1; i <= 6; ++i) { /* :409-419 */
            if (l[i] >= d) {
                l[i] -= d;
                aeva += d;
                d = 0.0;
            } else {
                aeva += l[i];
                d = pC * (d - l[i]);
                l[i] = 0.0;
            }
        }
    }

    out[0] = aeva / 1e3 * area / dt; /* :424 */

    /* five linear reservoirs, structure.py:427-450; "V += a - b" evaluates (a - b) first */
    double q;
    q = v_ove / sk;
    out[1] = q;
    v_ove += (of / 1e3 * area) - (q * dt);
    if (v_ove < 0.0) v_ove = 0.0;
    q = v_dra / sk;
    out[2] = q;
    v_dra += (df / 1e3 * area) - (q * dt);
    if (v_dra < 0.0) v_dra = 0.0;
    q = v_int / fk;
    out[3] = q;
    v_int += (inf / 1e3 * area) - (q * dt);
    if (v_int < 0.0) v_int = 0.0;
    q = v_sgw / gk;
    out[4] = q;
    v_sgw += (sh / 1e3 * area) - (q * dt);
    if (v_sgw < 0.0) v_sgw = 0.0;
    q = v_dgw / gk;
    out[5] = q;
    v_dgw += (dp / 1e3 * area) - (q * dt);
    if (v_dgw < 0.0) v_dgw = 0.0;

    out[7] = v_ove;
    out[8] = v_dra;
    out[9] = v_int;
    out[10] = v_sgw;
    out[11] = v_dgw;
    for (int i = 1; i <= 6; ++i)
        out[11 + i] = l[i] / 1e3 * area; /* :456-457 */

    /* river reservoir, structure.py:482-498; inflow summed left to right (:254) */
    const double qin = out[1] + out[2] + out[3] + out[4] + out[5];
    const double rk = p[9] * 3600.0;
    double v_riv = st[11];
    double q_riv = v_riv / rk;
    const double v_old = v_riv;
    const double tmp = v_old + (qin - q_riv) * dt;
    if (tmp < 0.0) {
        q_riv = 0.95 * (qin + v_old / dt);
        v_riv += (qin - q_riv) * dt;
    } else {
        v_riv = tmp;
    }
    out[6] = q_riv;
    out[18] = v_riv;
}

/* ---- numpy's pairwise summation (numpy/_core/src/umath/loops_utils.h.src, *_pairwise_sum),
 *      restated; stride in elements ---------------------------------------------------------- */
static double np_pairwise(const double *a, long n, long stride)
{
    if (n < 8) {
        double r = 0.0;
        for (long i = 0; i < n; ++i)
            r += a[i * stride];
        return r;
    }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; ++j)
            r[j] = a[j * stride];
        long i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j)
                r[j] += a[(i + j) * stride];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i)
            res += a[i * stride];
        return res;
    }
    long n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise(a, n2, stride) + np_pairwise(a + n2 * stride, n - n2, stride);
}

/* np.sum() of a whole contiguous array: the reduction is fed to the inner loop in buffers of 8192
 * elements, each pairwise-summed and added to the running total (measured against numpy 2.2.6 in
 * tests/golden/make_golden.py; the fancy-indexed operands of structure.py:191,194 are F-ordered). */
static double np_sum_flat(const double *a, long n)
{
    double r = 0.0;
    for (long i = 0; i < n; i += 8192) {
        long m = n - i < 8192 ? n - i : 8192;
        r = r + np_pairwise(a + i, m, 1);
    }
    return r;
}

/*
 * structure.py:149-197 run_all_steps.  rain/peva hold >= L values; initial/final are 19-vectors.
 * discharge holds L/gap values.  storage (nullable) receives the (L+1) x 19 table of :177.
 * Returns 0, or -1 on the conditions under which the reference raises (reshape of :190).
 */
int smart_oracle_all_steps(double area, double dt, long L, const double *rain, const double *peva,
                           const double *p, const double *initial, int report_type, long gap,
                           int pow_mode, int sum_mode, double *discharge, double *gw, double *final,
                           double *storage)
{
    if (gap <= 0 || L < 0)
        return -1;
    if (report_type == SMART_REPORT_SUMMARY && L % gap != 0)
        return -1; /* np.reshape(..., (-1, gap)) raises (:190) */
    if (report_type != SMART_REPORT_SUMMARY && report_type != SMART_REPORT_RAW)
        return -2;
    double *tab = storage ? storage : (double *)malloc(sizeof(double) * (size_t)(L + 1) * NVAR);
    if (!tab)
        return -3;
    memcpy(tab, initial, sizeof(double) * NVAR); /* :179 */
    for (long i = 1; i <= L; ++i)                /* :181-187 */
        smart_oracle_one_step(area, dt, rain[i - 1], peva[i - 1], p, tab + (i - 1) * NVAR + 7,
                              tab + i * NVAR, pow_mode);

    if (report_type == SMART_REPORT_SUMMARY) { /* :189-191 */
        const long R = L / gap;
        for (long r = 0; r < R; ++r) {
            const double *q = tab + (1 + r * gap) * NVAR + 6;
            double s;
            if (sum_mode == SMART_SUM_NUMPY || (sum_mode == SMART_SUM_GPU && gap <= 128)) {
                s = np_pairwise(q, gap, NVAR);
            } else {
                s = 0.0;
                for (long k = 0; k < gap; ++k)
                    s += q[k * NVAR];
            }
            discharge[r] = s / (double)gap; /* np.mean: add.reduce then true_divide by the count */
        }
        double num, den;
        if (sum_mode == SMART_SUM_NUMPY) {
            double *flat = (double *)malloc(sizeof(double) * (size_t)(L > 0 ? L : 1) * 5);
            if (!flat) {
                if (!storage) free(tab);
                return -3;
            }
            for (int c = 0; c < 5; ++c) /* F-ordered copy made by the fancy index [1,2,3,4,5] */
                for (long i = 0; i < L; ++i)
                    flat[c * L + i] = tab[(i + 1) * NVAR + 1 + c];
            den = np_sum_flat(flat, 5 * L);
            num = np_sum_flat(flat + 3 * L, 2 * L);
            free(flat);
        } else {
            num = 0.0;
            den = 0.0;
            for (long i = 1; i <= L; ++i) {
                const double *o = tab + i * NVAR;
                num += o[4] + o[5];
                den += (((o[1] + o[2]) + o[3]) + o[4]) + o[5];
            }
        }
        *gw = num / den;
    } else { /* :192-195: [::-gap][::-1] keeps rows L, L-gap, ... in increasing order */
        const long R = (L + gap - 1) / gap;
        double *flat = (double *)malloc(sizeof(double) * (size_t)(R > 0 ? R : 1) * 5);
        if (!flat) {
            if (!storage) free(tab);
            return -3;
        }
        for (long r = 0; r < R; ++r) {
            const long row = L - (R - 1 - r) * gap; /* index into tab (1-based rows of [1:]) */
            discharge[r] = tab[row * NVAR + 6];
            for (int c = 0; c < 5; ++c)
                flat[c * R + r] = tab[row * NVAR + 1 + c];
        }
        double num, den;
        if (sum_mode == SMART_SUM_NUMPY) {
            den = np_sum_flat(flat, 5 * R);
            num = np_sum_flat(flat + 3 * R, 2 * R);
        } else {
            num = 0.0;
            den = 0.0;
            for (long r = 0; r < R; ++r) {
                num += flat[3 * R + r] + flat[4 * R + r];
                den += (((flat[r] + flat[R + r]) + flat[2 * R + r]) + flat[3 * R + r]) + flat[4 * R + r];
            }
        }
        free(flat);
        *gw = num / den;
    }
    if (final)
        memcpy(final, tab + L * NVAR, sizeof(double) * NVAR); /* :197 */
    if (!storage)
        free(tab);
    return 0;
}

/* Initial 19-vector of structure.py:97-116 / :123-140.  extra = {aar, r-o_ratio, r-o_split[5]} or
 * NULL (reservoirs start empty). */
void smart_oracle_initial(double area, const double *p, const double *extra, double *init)
{
    for (int i = 0; i < NVAR; ++i)
        init[i] = 0.0;
    if (extra) {
        const double ro = extra[0] * extra[1];
        static const int kidx[5] = {6, 6, 7, 8, 8}; /* ove,dra->SK  int->FK  sgw,dgw->GK */
        for (int j = 0; j < 5; ++j)
            init[7 + j] = ro * extra[2 + j] / 1000 * area / 8766 * p[kidx[j]];
        init[18] = ro / 1000 * area / 8766 * p[9];
    }
    for (int i = 12; i <= 17; ++i)
        init[i] = (p[5] / 12) / 1000 * area;
}

/*
 * structure.py:30-146 run() for one sample: initial conditions, optional warm-up over the first
 * n_warm steps of the same forcing (:87-121), then the run proper (:143-146).
 * n_warm = int(warm_up_days * 86400 / dt) is computed by the caller.  final is nullable.
 * Returns 0; -1 for the reshape error; -4 when the warm-up is longer than the run (:90-95).
 */
static int run_with_scratch(double area, double dt, long n_steps, long n_warm, const double *rain,
                            const double *peva, const double *p, const double *extra, int report_type,
                            long gap, int pow_mode, int sum_mode, double *discharge, double *gw,
                            double *final, double *scratch)
{
    double init[NVAR];
    if (n_warm != 0) {
        if (n_warm > n_steps)
            return -4;
        double init_wu[NVAR], gw_wu;
        smart_oracle_initial(area, p, extra, init_wu);
        long Rw = (n_warm + gap - 1) / gap;
        double *dis_wu = (double *)malloc(sizeof(double) * (size_t)(Rw > 0 ? Rw : 1));
        if (!dis_wu)
            return -3;
        int rc = smart_oracle_all_steps(area, dt, n_warm, rain, peva, p, init_wu, report_type, gap,
                                        pow_mode, sum_mode, dis_wu, &gw_wu, init, scratch);
        free(dis_wu);
        if (rc)
            return rc;
    } else {
        smart_oracle_initial(area, p, extra, init);
    }
    return smart_oracle_all_steps(area, dt, n_steps, rain, peva, p, init, report_type, gap, pow_mode,
                                  sum_mode, discharge, gw, final, scratch);
}

int smart_oracle_run(double area, double dt, long n_steps, long n_warm, const double *rain,
                     const double *peva, const double *p, const double *extra, int report_type,
                     long gap, int pow_mode, int sum_mode, double *discharge, double *gw, double *final)
{
    return run_with_scratch(area, dt, n_steps, n_warm, rain, peva, p, extra, report_type, gap, pow_mode,
                            sum_mode, discharge, gw, final, NULL);
}

/*
 * The per-sample loop that spotpy's mc sampler drives (montecarlo.py:153-154,179-186), as a batch:
 * params[N][10] row-major, discharge[N][R] row-major (nullable), gw[N], final[N][19] (nullable).
 * OpenMP over samples; n_threads <= 0 uses the runtime default.  This is also what bench.py times
 * as the CPU baseline ("port").
 */
int smart_oracle_run_batch(long n_samples, double area, double dt, long n_steps, long n_warm,
                           const double *rain, const double *peva, const double *params,
                           const double *extra, int report_type, long gap, int pow_mode,
                           int sum_mode, double *discharge, double *gw, double *final, int n_threads)
{
    const long R = report_type == SMART_REPORT_SUMMARY ? n_steps / gap : (n_steps + gap - 1) / gap;
    int status = 0;
#ifdef _OPENMP
    if (n_threads > 0)
        omp_set_num_threads(n_threads);
#endif
#pragma omp parallel
    {
        /* one (L+1) x 19 table (structure.py:177) and one discharge row per thread, reused by its samples */
        double *table = (double *)malloc(sizeof(double) * (size_t)(n_steps + 1) * NVAR);
        double *row = (double *)malloc(sizeof(double) * (size_t)(R > 0 ? R : 1));
#pragma omp for schedule(dynamic, 1)
        for (long n = 0; n < n_samples; ++n) {
            double g = 0.0;
            int rc = (table && row) ? run_with_scratch(area, dt, n_steps, n_warm, rain, peva, params + n * 10,
                                                       extra, report_type, gap, pow_mode, sum_mode,
                                                       discharge ? discharge + n * R : row, &g,
                                                       final ? final + n * NVAR : NULL, table)
                                    : -3;
            gw[n] = g;
            if (rc) {
#pragma omp critical
                status = rc;
            }
        }
        free(table);
        free(row);
    }
    return status;
}

int smart_oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
