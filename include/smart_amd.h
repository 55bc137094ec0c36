/*
 * smart_amd.h -- C ABI of the MI355X-native SMART ensemble engine (libsmart_amd.so).
 *
 * This is the drop-in boundary for the hot path of ThibHlln/smartpy v0.2.2: the time loop
 * structure.run -> run_all_steps -> run_one_step (smartpy/structure.py:30-503).  The reference already
 * has a plug-in hook for exactly this path -- an optional extension module named `smartcpp`, looked up
 * at import time (structure.py:22-27) and used as smartcpp.allsteps (structure.py:56-62) or
 * smartcpp.onestep (structure.py:171-174).  Every entry point below states which reference interface
 * it stands in for.  INTEGRATION.md shows the ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *   - plain C types only; no torch / numpy types cross this boundary;
 *   - "device pointer" = memory the GPU can address (hipMalloc, or a torch.Tensor's data_ptr());
 *   - every function returns 0 on success or a negative SMART_E_* code, never throws; the text of the
 *     last error of the calling thread is available from smart_last_error();
 *   - buffers are owned by the caller, are not retained after the call returns and inputs are never
 *     written; launches are asynchronous on the given stream unless stated otherwise;
 *   - no global mutable state besides the per-thread error text: calls on different streams may run
 *     concurrently.
 *   - there is NO CPU fallback: without a usable HIP device every compute entry point fails with
 *     SMART_E_NO_DEVICE.
 *
 * Vector layouts (all IEEE fp64)
 *   parameters  p[10]  = T, C, H, D, S, Z, SK, FK, GK, RK                       (parameters.py:25)
 *   variables   v[19]  = Q_aeva, Q_ove, Q_dra, Q_int, Q_sgw, Q_dgw, Q_out,
 *                        V_ove, V_dra, V_int, V_sgw, V_dgw, V_ly1..V_ly6, V_river (structure.py:78-82)
 *   extra       x[7]   = aar, r-o_ratio, r-o_split[5]                          (structure.py:100-112)
 *   objectives  o[8]   = NSE, KGE, KGEc, KGEa, KGEb, PBias, RMSE, GW           (montecarlo.py:71-74)
 */
#ifndef SMART_AMD_H
#define SMART_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SMART_AMD_ABI_VERSION 7

/* report_type, as structure.py:65-70 maps report='summary' / 'raw' */
#define SMART_REPORT_SUMMARY 1
#define SMART_REPORT_RAW 2

/* math_mode */
#define SMART_MATH_LITERAL 0 /* the reference's operation order, IEEE division, no FMA contraction:     */
                             /* bit-identical to the CPU oracle configured with the product chain for   */
                             /* s'**i (the only place where CPython calls libm)                         */
#define SMART_MATH_FAST 1    /* reciprocals hoisted out of the time loop, soil layers kept in mm,       */
                             /* reservoirs kept as outflows, FMA: <= 1e-9 relative on discharge;        */
                             /* wavefronts holding a sample with delta_sec / (k * 3600) > 2 (where the  */
                             /* reference's explicit update amplifies rounding differences) run the     */
                             /* literal arithmetic instead, and so do rows with a NaN or an infinite    */
                             /* parameter or one that is none (D or H outside [0, 1], a k <= 0,         */
                             /* T < 0.2, Z outside [1 mm, 1 m]: smart_fast_model.h wave_class).         */
                             /* The FORCING must be finite in this mode: what the                       */
                             /* reference's branches make of a NaN in it (structure.py:359, :409-419)   */
                             /* only SMART_MATH_LITERAL reproduces -- a fast launch that meets one      */
                             /* raises SMART_STATUS_NONFINITE_FORCING                                   */

/* error codes; the reference raises Exception at the cited places */
#define SMART_OK 0
#define SMART_E_NULL (-1)        /* a required pointer is NULL                                          */
#define SMART_E_SIZE (-2)        /* a size is <= 0 or inconsistent                                      */
#define SMART_E_REPORT_TYPE (-3) /* unknown report type                          (structure.py:69-70)   */
#define SMART_E_WARMUP (-4)      /* warm-up longer than the simulation           (structure.py:90-95)   */
#define SMART_E_GAP (-5)         /* summary report: length (or warm-up) is not a multiple of the gap    */
                                 /* -- np.reshape raises                         (structure.py:190)     */
#define SMART_E_NO_DEVICE (-6)   /* no HIP device / HIP runtime error (text in smart_last_error)        */
#define SMART_E_MODE (-7)        /* unknown math mode                                                   */
#define SMART_E_IO (-8)          /* a database file could not be opened / written                       */

/* plan bits (smart_plan_ensemble, SmartEnsemble.plan): what the rows and the forcing of a call need.  The fast mode
 * is a family of kernels, one per arithmetic class of a block of 64 parameter rows and per kind of forcing; a plan
 * lets a launch skip the kernels that would find nothing to do. */
#define SMART_PLAN_CLASS_REGULAR 0x01    /* rows with every k*3600 >= delta_sec, 0 <= S <= 0.5, C >= 0, Z > 0    */
#define SMART_PLAN_CLASS_STIFF 0x02      /* some k*3600 < delta_sec: clamps (structure.py:429-450) and the river's */
                                         /* 95 % rule (:492-496) are reachable                                     */
#define SMART_PLAN_CLASS_GUARD 0x04      /* S, C or Z outside those ranges: the leak guards (:383,390,397) matter  */
#define SMART_PLAN_CLASS_ILLCOND 0x08    /* delta_sec / (RK*3600) > 2 (the river), or a parameter that is none     */
                                         /* (SMART_MATH_FAST above): run in the literal arithmetic                 */
#define SMART_PLAN_FORCING_PIECEWISE 0x10 /* a catchment whose forcing is constant within every report interval    */
#define SMART_PLAN_FORCING_VARYING 0x20   /* a catchment whose forcing varies from step to step                    */
#define SMART_PLAN_FORCING_RUNS 0x80      /* a catchment whose forcing is constant over runs of k steps, k >= 2 a  */
                                          /* divisor of the report gap smaller than it (6-hourly data in an hourly  */
                                          /* run with daily reports, timeframe.py:167-186): interval engine per run */
#define SMART_PLAN_ROWS_ORDERED 0x40       /* set by the caller (not by smart_plan_ensemble): neighbouring rows behave    */
                                           /* alike -- ordered by T, then by S*Z, as smartpy_amd/engine.py does when no   */
                                           /* discharge matrix is stored.  Wave-uniform early exits then pay off sooner.  */
#define SMART_PLAN_VALID 0x100
/* ABI 7: bits 12..30 of a plan from smart_plan_ensemble hold the NUMBER of blocks of 64 rows that take the literal      */
/* arithmetic (SMART_PLAN_CLASS_ILLCOND), saturating; 0 in a plan put together by hand = not counted (the launch then    */
/* reckons with every block of the call).  The launch picks the form of that kernel from it: see literal_form.           */
#define SMART_PLAN_ILLCOND_BLOCKS_SHIFT 12
#define SMART_PLAN_ILLCOND_BLOCKS_MAX 0x7ffff

/* literal_form (ABI 7): how the rows of SMART_PLAN_CLASS_ILLCOND are laid over the wavefronts.  The same arithmetic,   */
/* the same bits either way (structure.py:267-503 in the reference's operation order).                                  */
#define SMART_LITERAL_FORM_AUTO 0  /* from the load: rows while the wavefronts of the call -- 16 per class-3 block, one    */
                                   /* per other block -- fit into two rounds over the chip's SIMDs, lanes beyond         */
#define SMART_LITERAL_FORM_ROWS 1  /* one sample per DPP row of 16 lanes, four per wavefront: a third of the             */
                                   /* instructions per step -- the latency form (few rows: config 2, the smartcpp hook)  */
#define SMART_LITERAL_FORM_LANES 2 /* one sample per lane, 64 per wavefront -- the throughput form (large daily          */
                                   /* ensembles: a seventh of the SIMD cycles per sample-step)                           */

/* status bits of a finished launch (smart_launch_status) */
#define SMART_STATUS_SLICE_TIMEOUT 0x1 /* a time slice gave up waiting for its predecessor: the block's outputs are */
                                       /* NaN from that slice on; repeat the launch with time_slices = 1            */
#define SMART_STATUS_STALE_PLAN 0x2    /* the plan did not cover a class / kind of forcing met on the device: those  */
                                       /* rows were NOT computed; repeat the launch with plan = 0                    */
#define SMART_STATUS_NONFINITE_FORCING 0x4 /* SMART_MATH_FAST met a NaN or an infinity in the forcing: its outputs are */
                                       /* not the reference's; repeat the launch with math_mode = SMART_MATH_LITERAL  */

/*
 * One ensemble launch = the whole per-sample loop that spotpy's sampler drives
 * (montecarlo.py:153-154 -> :179-186 -> smart.py:204-207 -> structure.py:30-146), for n_samples
 * parameter sets on each of n_catchments catchments.  One wavefront lane advances one sample; the time
 * loop (warm-up then simulation) runs inside the kernel with parameters and the 12 states in registers.
 *
 * All pointers are device pointers.
 */
typedef struct SmartEnsemble {
    /* ---- sizes -------------------------------------------------------------------------------- */
    int64_t n_catchments; /* C >= 1                                                                 */
    int64_t n_samples;    /* N >= 1 parameter sets per catchment                                    */
    int64_t n_steps;      /* T = len(timeseries) - 1                          (structure.py:73)      */
    int64_t n_warm;       /* W = int(warm_up_days * 86400 / delta_sec), 0 = none (structure.py:87-88); */
                          /* the warm-up replays forcing[0 .. W)              (structure.py:118-121) */
    int64_t report_gap;   /* g = T // R                                       (structure.py:75)      */
    int32_t report_type;  /* SMART_REPORT_*                                                          */
    int32_t math_mode;    /* SMART_MATH_*                                                            */
    double delta_sec;     /* simulation time step in seconds                  (structure.py:74)      */

    /* ---- inputs ------------------------------------------------------------------------------- */
    const double *area_m2; /* [C]                                                                    */
    const double *forcing; /* [C][T][2]: rain, peva in mm per step, interleaved (smart.py:141-142)   */
    const double *params;  /* [N][10] row-major, exactly the matrix lhs.py:114 produces; or [C][N][10] */
    int64_t params_catchment_stride; /* 0: every catchment runs the same N rows; else N*10          */
    const double *extra;   /* [C][7] educated guess of the initial reservoirs, or NULL: start empty   */
                           /*                                                 (structure.py:97-140)  */
    const double *initial; /* [C][N][12] states to start from instead (chained runs), or NULL;        */
                           /* used as the warm-up's start when n_warm != 0                           */
    const double *obs;     /* [C][R] observed discharge, NaN = missing (smart.py:143), or NULL        */
    const double *gw_obs;  /* [C] groundwater constraint (inout.py:134-139), NaN = none, or NULL      */

    /* ---- outputs (each nullable except gw) ----------------------------------------------------- */
    double *discharge;     /* [C][R][discharge_ld] sample-minor, so that a wavefront stores 512       */
                           /* contiguous bytes per report step; element (c, r, n) = what              */
                           /* SMART.simulate(row n)[0][r] returns                   (smart.py:208)   */
    int64_t discharge_ld;  /* >= N                                                                   */
    double *gw;            /* [C][N] groundwater contribution to runoff (structure.py:191,194-195)   */
    double *objfn;         /* [C][N][8] objective functions (montecarlo.py:193-209); needs obs;       */
                           /* column 7 (GW) is NaN where gw_obs is absent                            */
    double *final_vars;    /* [C][N][19] last row of the storage table        (structure.py:197)     */
    void *workspace;       /* device scratch of workspace_bytes bytes: observation statistics when    */
                           /* objfn != NULL ([C][8 + R] doubles, required), and the hand-over buffer   */
                           /* of a time-sliced launch (optional: without room for it the launch is     */
                           /* not sliced), and the code words of the step loop's pair blocks (optional:  */
                           /* without them it dispatches step by step).  smart_workspace_bytes() tells   */
                           /* how much all of them need.                                               */
    int64_t workspace_bytes;

    void *stream;          /* hipStream_t; NULL = the default stream                                 */

    /* ---- launch control (ABI 3; zero = the library decides) ------------------------------------- */
    int32_t time_slices;   /* 0: the library's choice (slices the time axis of launches with more blocks  */
                           /* of 64 samples than SIMDs); 1: never slice; n > 1: n slices                   */
    int32_t plan;          /* SMART_PLAN_* bits from smart_plan_ensemble for these params / forcing /      */
                           /* sizes, or 0: launch every kernel the call could need                         */
    /* ---- ABI 7 ---------------------------------------------------------------------------------- */
    int32_t literal_form;  /* SMART_LITERAL_FORM_*: 0 = chosen from the number of class-3 blocks in the plan */
    int32_t reserved0;     /* must be 0                                                                    */
} SmartEnsemble;

/* Number of report steps R for a run (structure.py:190 / :193): T // g (summary), ceil(T / g) (raw). */
int64_t smart_n_reports(int64_t n_steps, int64_t report_gap, int32_t report_type);

/* Validate and launch.  Asynchronous on e->stream.  Stands in for the spotpy repetition loop over
 * MonteCarlo.simulation + MonteCarlo.objectivefunction (montecarlo.py:179-209). */
int smart_run_ensemble_hip(const SmartEnsemble *e);

/* Validation only (no device needed): the checks smart_run_ensemble_hip performs before launching. */
int smart_check_ensemble(const SmartEnsemble *e);

/* Bytes of device scratch the call wants in e->workspace for these sizes and outputs (pointers are not read):
 * a header (status word, counters, per-catchment forcing flags), the observation statistics if e->objfn is set,
 * plus -- on a machine with a HIP device -- the hand-over buffer of the time-sliced launch the library would
 * choose, plus, for fast summary / raw runs over whole report intervals, 8 bytes per four time steps and catchment
 * (the kinds of the steps, worked out once per launch; 68 bytes per two steps where the gap is no multiple of four).
 * The library allocates nothing itself: the caller owns every buffer, which also lets the call be captured into a HIP
 * graph.  A launch without a workspace runs unsliced and reports no status. */
int64_t smart_workspace_bytes(const SmartEnsemble *e);

/* Classify the parameter rows and the forcing of a SMART_MATH_FAST call on the device (two small kernels on
 * e->stream, then SYNCHRONOUS: the answer is copied back): *plan = SMART_PLAN_VALID | the SMART_PLAN_* bits present.
 * Valid for as long as params, forcing, delta_sec, report_gap and the sizes do not change.  Needs e->workspace. */
int smart_plan_ensemble(const SmartEnsemble *e, int32_t *plan);

/* Status word of the last launch that used e->workspace (SMART_STATUS_* bits, 0 = clean).  SYNCHRONOUS: waits for
 * e->stream.  0 without a workspace. */
int smart_launch_status(const SmartEnsemble *e, int32_t *status);

/* Which kernels smart_run_ensemble_hip would launch for *e on the current device, as text for logs and benchmark
 * lines: e.g. "smart_fast_intervals[16 slices x 1563 blocks, 2 resident per SIMD]"; the literal rows of a fast call
 * with the form chosen for them: "smart_fast_illcond_lanes[1813 of 15625 blocks, one sample per lane]" (ABI 7).
 * Launches nothing. */
int smart_describe_launch(const SmartEnsemble *e, char *text, int64_t len);

/*
 * smartcpp.allsteps -- same arguments and results as run_all_steps (structure.py:149-152,197).
 * HOST pointers; synchronous (copies in, runs one sample on the GPU, copies out).
 *   nd_rain / nd_peva hold at least length_simu values (the warm-up call passes the full series with a
 *   shorter length, structure.py:118-121); nd_parameters[10]; nd_initial[19];
 *   discharge[smart_n_reports(length_simu, report_gap, report_type)]; *groundwater_component;
 *   final_vars[19].
 * Arithmetic: SMART_MATH_LITERAL (the reference's operation order; bit-identical to the ensemble's literal mode) unless
 * the environment says SMART_ALLSTEPS_MATH=fast (the fast kernels for the one sample: <= 1e-9 relative, a tenth of the
 * time).  The library keeps its device buffers and the uploaded series between calls (ABI v5): the reference calls this
 * twice per SMART.simulate() with the same series (structure.py:118-121,143-146), a calibration loop thousands of
 * times -- a call whose series starts with what is on the device uploads only what lies beyond it.  One caller at a
 * time (serialised inside).
 */
int smart_allsteps_hip(double area_m2, double delta_sec, int64_t length_simu, const double *nd_rain,
                       const double *nd_peva, const double *nd_parameters, const double *nd_initial,
                       int32_t report_type, int64_t report_gap, double *discharge,
                       double *groundwater_component, double *final_vars);

/* Counters of the smartcpp.allsteps stand-in, for tests and logs: counters[0..n) of { calls, device allocations made,
 * bytes of forcing uploaded, calls served in fast arithmetic, planning passes made for them (ABI 6: one per series,
 * length and report, not one per call) } since the library was loaded (n <= 5). */
int smart_hook_counters(int64_t *counters, int64_t n);

/*
 * smartcpp.onestep -- run_one_step (structure.py:200-264): 2 constants, 2 forcings, 10 parameters and
 * 12 states in, the 19-vector out.  HOST pointers; synchronous; n independent steps per call
 * (n = 1 reproduces the reference's call at structure.py:182-187).
 *   in[n][26]  = area, dt, rain, peva, T..RK, V_ove..V_river ;  out[n][19]
 */
int smart_onestep_hip(int64_t n, const double *in, double *out);

/*
 * run_one_step_river (structure.py:461-503) on its own: the river reservoir with its 95 % rule.  HOST pointers;
 * synchronous; n independent steps.  in[n][4] = time_gap_sec, r_in_q_riv, r_p_rk [hours], r_s_v_riv ;
 * out[n][2] = r_out_q_riv, r_s_v_riv.  (run_one_step_catchment, :267-458, is the first six outputs and the eleven
 * catchment states of smart_onestep_hip: the river does not feed back.)
 */
int smart_river_step_hip(int64_t n, const double *in, double *out);

/*
 * Objective functions of an existing discharge matrix (montecarlo.py:193-209 applied to every sample);
 * the matrix is read once (moments about the observation mean).  Device pointers; asynchronous on stream.
 *   sim[R][ld] sample-minor (the layout smart_run_ensemble_hip writes), obs[R] (NaN = missing),
 *   gw_sim[N] and gw_obs: pass NULL / NaN to skip the GW column; objfn[N][8].
 */
int smart_objfn_hip(int64_t n_samples, int64_t n_reports, const double *sim, int64_t ld, const double *obs,
                    const double *gw_sim, double gw_obs, double *objfn, void *stream);

/*
 * Sampling database, CSV flavour -- the rows MonteCarlo.save writes one by one (montecarlo.py:211-231): every value
 * cast to float32 and printed '%.6e', comma separated, one '\n'-terminated line per sample.  HOST pointers, no
 * device involved.  Appends n_rows lines of n_cols values (row-major float32 table: objective functions, parameters,
 * optionally the simulated series -- the caller writes the header line, montecarlo.py:122-127) to `path`.
 * n_threads <= 0: one per core, at most 16.  Characters identical to Python's '%.6e' % numpy.float32(x).
 */
int smart_db_append_rows(const char *path, const float *table, int64_t n_rows, int64_t n_cols, int32_t n_threads);

/*
 * ... and reading them back: GLUE / Best / Total take their sample from an LHS database (montecarlo.py:233-262,
 * DictReader + numpy.array(rows, dtype=float32)).  `text` = the file's bytes after the header line, `len` of them;
 * every line holds n_cols values; the n_use columns listed in `cols` (indices into the header) are parsed
 * (text -> correctly rounded double -> float32, as numpy does) into out[row][0..n_use).  Returns the number of rows
 * (<= max_rows) or a negative SMART_E_* code.
 */
int64_t smart_db_parse_rows(const char *text, int64_t len, int64_t n_cols, const int32_t *cols, int32_t n_use,
                            float *out, int64_t max_rows, int32_t n_threads);

/* The arithmetic class of ONE parameter row -- 0 regular, 1 stiff, 2 guard, 3 ill-conditioned / literal: the c of the
 * SMART_PLAN_CLASS_* bit (1 << c) its kernel answers to -- worked out on the host with the rules the kernels apply on
 * the device (smart_fast_model.h: wave_class).  params[10] as lhs.py:114 lays them out; initial12: the twelve initial
 * states of the row, or NULL (the reference's half-full soil / educated guess, structure.py:97-140); area_m2 is read with
 * initial12 only.  Pure host code, no device needed.  smart_allsteps_hip classifies its one row with it (ABI 6); a
 * caller that launches one row at a time can fill SmartEnsemble.plan the same way:
 *   plan = SMART_PLAN_VALID | (forcing bits of an earlier smart_plan_ensemble) | (1 << smart_row_class(...)).
 * The device stays authoritative: a row this call misjudges meets no kernel and raises SMART_STATUS_STALE_PLAN. */
int smart_row_class(const double *params, double delta_sec, const double *initial12, double area_m2);

/* Device bookkeeping */
int smart_device_count(void);           /* number of visible HIP devices (0 if none / no driver)      */
int smart_abi_version(void);            /* SMART_AMD_ABI_VERSION the library was built with           */
const char *smart_build_info(void);     /* the compiler that built the library, the ABI, and the stamp of the build's code lint:
                                         * "hipcc <HIP version> | clang <version> | gfx950 | ABI 7 | SMART_LINT_STAMP=pairs-ok"
                                         * (smartpy_amd.build writes the stamp into the file it has linted; "unchecked": a
                                         * build by another route -- the step loops then run without computed jumps unless
                                         * the environment says SMART_PAIR_BLOCKS=1) */
const char *smart_last_error(void);     /* text of the calling thread's last error ("" if none)       */

#ifdef __cplusplus
}
#endif
#endif /* SMART_AMD_H */
