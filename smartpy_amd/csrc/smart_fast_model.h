// smart_fast_model.h -- SMART_MATH_FAST: the same recurrence (structure.py:267-503) re-expressed for the
// fp64 vector ALU of gfx950.  There is no contraction in this model (a strict, branchy recurrence per
// sample), so no MFMA; the binding roof is fp64 VALU issue, and the per-step instruction count is what
// this file minimises:
//
//   * every division by a per-sample constant (area, 1e3, Z, k*3600, i, dt, gap) becomes a multiplication
//     by a reciprocal computed once before the time loop (the literal form has 39 / 26 IEEE divisions per
//     wet / dry step, ~10 instructions each);
//   * soil layers stay in mm: the 12 conversions V/area*1e3 <-> lvl/1e3*area of structure.py:339-347 and
//     :456-457 disappear;
//   * each linear reservoir is carried as its outflow U = V / k (m3/s):  U' = U*(1 - dt/k) + x*(area/1e3/k),
//     one FMA + one multiply, and the five outflows needed by the river and by the groundwater ratio are
//     the state itself;
//   * s'**i by multiplication, leaks as FMA pairs, top-down filling as min/sub chains;
//   * the clamps V < 0 -> 0 (structure.py:429-450) and the river's 95 % rule (:492-496) can only fire when
//     k*3600 < dt, and the "leak < level" guards (:383,390,397) only when s' >= 1: each wavefront tests its
//     own 64 parameter rows once, before the time loop, and takes the STIFF / GUARD instantiation of the
//     loop only if one of its lanes needs it (a wave-uniform branch outside the loop);
//   * reservoirs that share a time constant are merged and the two sums behind the groundwater ratio come
//     from mass balances (MERGE, below) when the caller does not ask for the final state vector;
//   * a wavefront holding a sample with dt / RK > 2, where the reference's own river update amplifies rounding
//     differences, runs the literal model (smart_literal_model.h) instead;
//   * summary reports over forcing that is constant within the report interval (daily data spread over the hours,
//     the reference's own input pipeline) advance one interval at a time (run_ensemble_merged in smart_device.h):
//     a dry interval is ONE evaporation step with gap times the demand plus a precomputed linear routing map
//     (dry_interval), a wet interval is `gap` straight-line wet steps whose leak amounts come from mass balance
//     (wet_balance: 74 vector instructions per step against 117 in the step loop);
//   * the same engine advances forcing that is constant over shorter runs (k steps, k a divisor of the report gap:
//     6-hourly data in an hourly run) a run at a time, the report mean accumulating across runs (smart_fast_runs);
//     runs without rain are classified on the scalar unit (every lane dry, or -- no evaporation either -- calm: wet
//     steps with nothing to fill);
//   * forcing that varies from step to step takes the step loop, written at the instruction level: three asm arms
//     (dry / calm / rain) picked per step on the scalar unit, threaded through chunks of four steps
//     (smart_fast_arms.h); the wet interval of the straight-line interval kernel is an asm loop as well;
//   * launches with more blocks of 64 samples than SIMDs are time-sliced (smart_device.h) so that the hardware
//     dispatcher evens out what a whole-run-per-wavefront mapping leaves idle.
//
// Rounding differs from the reference at the 1e-16 level per operation; the recurrence is dissipative (for
// dt / k <= 2), so the discharge stays within ~1e-12 relative of the reference (measured 1.5e-12 at most over 256 LHS
// rows x 10 years hourly, median 2.5e-14; gate in tests: 1e-9; contract: 1e-6).
//
// Tuning knobs (macros) are kept so that tools/ab_variants.sh can A/B them on one box; the defaults are the
// measured winners.  Tried and not kept: see DESIGN.md section 4.1.
#pragma once

#include "smart_device.h"
#include "smart_fast_arms.h"

#ifndef SMART_FAST_EARLY_EXIT
#define SMART_FAST_EARLY_EXIT 1
#endif
#ifndef SMART_FAST_DRY_EXIT2
#define SMART_FAST_DRY_EXIT2 1
#endif
#ifndef SMART_FAST_INTERVALS
#define SMART_FAST_INTERVALS 1
#endif
#ifndef SMART_FAST_BALANCE_SUMS
#define SMART_FAST_BALANCE_SUMS 1
#endif
#ifndef SMART_FAST_LEAK_BALANCE
#define SMART_FAST_LEAK_BALANCE 1
#endif
#ifndef SMART_FILL_EXITS
#define SMART_FILL_EXITS 2 // exits after the first and after the second layer of the filling cascade (kExits kernels)
#endif
#ifndef SMART_WET_UNROLL
#define SMART_WET_UNROLL 1
#endif


namespace smart {

// MERGE: reservoirs that share a time constant are linear and un-clamped in the regular case, so the pair
// (overland, drain) [k = SK] and the pair (shallow, deep groundwater) [k = GK] can each be carried as ONE
// outflow: (Ua + Ub)' = (Ua + Ub)*dec + (xa + xb)*cq.  Only their sums enter the river inflow (structure.py:254)
// and the groundwater ratio (:191).  Used when the caller does not ask for the final state vector.
// EXITS: wave-uniform early exits inside the interval engine's cascades (see kExits below).
// SPLIT (with MERGE): the drain and the deep-groundwater reservoir are carried next to the two merged totals, for
// callers that ask for the final state vector; the discharge arithmetic is that of the merged form, bit for bit.
template <bool STIFF, bool GUARD, bool MERGE = false, bool EXITS = true, bool SPLIT = false>
struct FastModel {
    static_assert(!(MERGE && STIFF), "the clamps of structure.py:429-450 act on each reservoir separately");
    static_assert(MERGE || !SPLIT, "SPLIT refines the merged form");
    static constexpr bool kExactDivide = false;
    static constexpr bool kSplit = SPLIT;

    // per-sample constants
    double pT, pC, pD, hz, sz, z;
    double dec_s, dec_f, dec_g, cq_s, cq_f, cq_g; // 1 - dt/k and area/1e3/k per routing constant
    double a_r, inv_a_r;                          // dt / rk and its reciprocal
    double a_s, a_f, a_g;                         // dt / k
    double car_s, car_f, car_g, om_ar;            // merged variant: a_r * area/1e3/k per reservoir, 1 - a_r
    double k_s, k_f, k_g, k_r, mm_to_m3;          // only to convert the states back at the end
    // states
    double l0, l1, l2, l3, l4, l5; // soil layers, mm
    double u_ove, u_dra, u_int, u_sgw, u_dgw, u_riv;
    // per-step results
    double q_out, q_in, q_gw;

    __device__ void setup(double area, double dt, const double *p)
    {
        pT = p[0];
        pC = p[1];
        pD = p[3];
        const double inv_z = 1.0 / p[5];
        hz = p[2] * inv_z;
        sz = p[4] * inv_z;
        z = p[5] / 6.0;
        k_s = p[6] * 3600.0;
        k_f = p[7] * 3600.0;
        k_g = p[8] * 3600.0;
        k_r = p[9] * 3600.0;
        mm_to_m3 = area / 1e3;
        const double ik_s = 1.0 / k_s, ik_f = 1.0 / k_f, ik_g = 1.0 / k_g;
        dec_s = 1.0 - dt * ik_s;
        dec_f = 1.0 - dt * ik_f;
        dec_g = 1.0 - dt * ik_g;
        cq_s = mm_to_m3 * ik_s;
        cq_f = mm_to_m3 * ik_f;
        cq_g = mm_to_m3 * ik_g;
        a_r = dt / k_r;
        a_s = dt * ik_s;
        a_f = dt * ik_f;
        a_g = dt * ik_g;
        car_s = a_r * cq_s;
        car_f = a_r * cq_f;
        car_g = a_r * cq_g;
        om_ar = 1.0 - a_r;
        inv_a_r = k_r / dt;
    }

    __device__ void set_states(const double *st)
    {
        const double m3_to_mm = 1.0 / mm_to_m3;
        if (MERGE) { // merged reservoirs are carried as volumes in mm: y' = y (1 - dt/k) + x, outflow = y area/1e3/k
            u_ove = (st[0] + st[1]) * m3_to_mm;
            u_int = st[2] * m3_to_mm;
            u_sgw = (st[3] + st[4]) * m3_to_mm;
            u_dra = SPLIT ? st[1] * m3_to_mm : 0.0;
            u_dgw = SPLIT ? st[4] * m3_to_mm : 0.0;
        } else {     // the others as outflows U = V / k
            u_ove = st[0] / k_s;
            u_dra = st[1] / k_s;
            u_int = st[2] / k_f;
            u_sgw = st[3] / k_g;
            u_dgw = st[4] / k_g;
        }
        l0 = st[5] * m3_to_mm;
        l1 = st[6] * m3_to_mm;
        l2 = st[7] * m3_to_mm;
        l3 = st[8] * m3_to_mm;
        l4 = st[9] * m3_to_mm;
        l5 = st[10] * m3_to_mm;
        u_riv = st[11] / k_r;
        note_capacity();
    }

    // Wave-uniform early exits in the FILLING cascade assume that a layer below the one just filled cannot overflow
    // by itself.  True for every state the model produces with C >= 0 -- but a caller's initial state may hold a
    // layer above its capacity z = Z / 6 (the reference then spills it downwards at the next wet step,
    // structure.py:367-374, even on a step whose own excess is used up in the top layer), and with C < 0 (GUARD) the
    // evaporation cascade itself pushes layers above capacity.  over_mask is all ones in those cases and ORed into
    // every exit's ballot, so that such a wavefront walks all six layers.
    unsigned long long over_mask;

    __device__ void note_capacity()
    {
        const bool over = l0 > z || l1 > z || l2 > z || l3 > z || l4 > z || l5 > z;
        over_mask = GUARD || __builtin_amdgcn_ballot_w64(over) != 0 ? ~0ull : 0ull;
    }

    __device__ __forceinline__ bool excess_left(double ex) const
    {
        return (__builtin_amdgcn_ballot_w64(ex > 0.0) | over_mask) != 0;
    }

    // The seven outputs of the LAST step (structure.py:197 returns the whole last row of the storage table) are not
    // carried through the loop: nothing reads them before the end (run() returns [0:2] of run_all_steps, the warm-up
    // hand-over uses the states only, :118-121,143-146,182-187).  The loop skeletons call flows_of_next_step() once,
    // on the state the last step starts from, and hand the result to get_vars().
    static constexpr bool kTracksOutputs = false;

    __device__ void flows_of_next_step(double dt, double rain_in, double peva_in, double *v) const
    {
        // outflows of a step are the reservoir states at its start (structure.py:427, :487)
        if (MERGE) { // totals and (SPLIT) their drain / deep parts, volumes in mm
            v[1] = (u_ove - u_dra) * cq_s;
            v[2] = u_dra * cq_s;
            v[3] = u_int * cq_f;
            v[4] = (u_sgw - u_dgw) * cq_g;
            v[5] = u_dgw * cq_g;
        } else {
            v[1] = u_ove;
            v[2] = u_dra;
            v[3] = u_int;
            v[4] = u_sgw;
            v[5] = u_dgw;
        }
        const double inflow = ((v[1] + v[2]) + v[3]) + (v[4] + v[5]);
        double q_r = u_riv;
        if (STIFF && fma(inflow - u_riv, a_r, u_riv) < 0.0) // 95 % rule, :492-496
            q_r = 0.95 * fma(u_riv, inv_a_r, inflow);
        v[6] = q_r;
        // actual evaporation (:361, :401-419): the potential rate on a wet step; on a dry one the (scaled) rain plus
        // what the cascade takes from the layers
        const double ex = excess(rain_in, peva_in);
        double aeva = peva_in;
        if (!(ex >= 0.0)) {
            const double lv[6] = {l0, l1, l2, l3, l4, l5};
            double d = -ex;
            aeva = rain_in * pT;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const bool enough = lv[i] >= d;
                aeva += enough ? d : lv[i];
                d = enough ? 0.0 : pC * (d - lv[i]);
            }
        }
        v[0] = aeva * mm_to_m3 / dt;
    }

    // `flows`: the seven outputs from flows_of_next_step(), or null (NaN: the caller did not capture them)
    __device__ void get_vars(double *v, const double *flows) const
    {
#pragma unroll
        for (int i = 0; i < 7; ++i)
            v[i] = flows ? flows[i] : quiet_nan();
        if (MERGE) { // totals and (SPLIT) their drain / deep parts, in mm; without SPLIT the pairs stay merged
            v[7] = (u_ove - u_dra) * mm_to_m3;
            v[8] = u_dra * mm_to_m3;
            v[9] = u_int * mm_to_m3;
            v[10] = (u_sgw - u_dgw) * mm_to_m3;
            v[11] = u_dgw * mm_to_m3;
        } else {
            v[7] = u_ove * k_s;
            v[8] = u_dra * k_s;
            v[9] = u_int * k_f;
            v[10] = u_sgw * k_g;
            v[11] = u_dgw * k_g;
        }
        v[12] = l0 * mm_to_m3;
        v[13] = l1 * mm_to_m3;
        v[14] = l2 * mm_to_m3;
        v[15] = l3 * mm_to_m3;
        v[16] = l4 * mm_to_m3;
        v[17] = l5 * mm_to_m3;
        v[18] = u_riv * k_r;
    }

    // top-down filling of one layer (structure.py:367-374): a = min(ex, space)
    __device__ static void fill(double &l, double &ex, double z)
    {
        const double a = fmin(ex, z - l);
        l += a;
        ex -= a;
    }

    // one leak of one layer (structure.py:381-399).  With 0 <= s' < 1 the guard "leak < level" is false
    // only for an empty layer, where the unguarded update is a no-op.
    __device__ static void leak1(double &l, double p, double &flow)
    {
        if (GUARD) { // value selects, not control flow: keeps every state in registers
            const double lk = l * p;
            const bool on = lk < l;
            flow = on ? flow + lk : flow;
            l = on ? l - lk : l;
        } else {
            flow = fma(l, p, flow);
            l = fma(-l, p, l);
        }
    }

    __device__ static void leak(double &l, double pa, double pb, double pc, double &inf, double &sh, double &dp)
    {
        leak1(l, pa, inf);
        leak1(l, pb, sh);
        leak1(l, pc, dp);
    }

    // "if V < 0: V = 0" of structure.py:429-450; can only fire when k*3600 < dt
    __device__ static double clamp(double u) { return STIFF ? fmax(u, 0.0) : u; }

    // evaporation demand taken from one layer (structure.py:409-419); needs C >= 0 (else GUARD)
    __device__ static void dry(double &l, double &d, double c)
    {
        if (GUARD) {
            const bool enough = l >= d;
            const double ln = enough ? l - d : 0.0;
            const double dn = enough ? 0.0 : c * (d - l);
            l = ln;
            d = dn;
        } else {
            const double t = d - l;
            l = fmax(-t, 0.0);
            d = fmax(c * t, 0.0);
        }
    }

    // rain excess of a step (structure.py:353-355); evaluated a few steps ahead by time_loop()
    __device__ double excess(double rain_in, double peva_in) const { return fma(rain_in, pT, -peva_in); }

    __device__ double layer_sum() const { return ((l0 + l1) + (l2 + l3)) + (l4 + l5); }

    // y = y * a + x with the result in y's own register (v_fma_f64, dst = src0).  hipcc prefers v_fmac_f64 into the
    // register of the dying addend and then pays a v_mov_b64 per loop-carried value at the end of every step.
    __device__ static void fma_in_place(double &y, double a, double x)
    {
        asm("v_fma_f64 %0, %0, %1, %2" : "+v"(y) : "v"(a), "v"(x));
    }

    // sum += y; y = y * a + x -- as one unit, so that the scheduler cannot move the update ahead of the sum and then
    // keep y's old value alive in a second register (a v_mov_b64 per step on the loop's back-edge)
    __device__ static void sum_then_fma_in_place(double &sum, double &y, double a, double x)
    {
        asm("v_add_f64 %0, %0, %1\n\tv_fma_f64 %1, %1, %2, %3" : "+v"(sum), "+v"(y) : "v"(a), "v"(x));
    }


    // wet branch of structure.py:359-399 for the lanes that are active
    __device__ __forceinline__ void wet_lanes(double ex)
    {
        if (kLeakBalance) {
            double tot = layer_sum();
            wet_balance(ex, ex * hz, tot);
            return;
        }
        const double tot = layer_sum();
        const double hp = hz * tot;
        const double s1 = sz * tot;
        const double of = hp * ex;
        ex = fma(-hp, ex, ex);
        fill(l0, ex, z);
#if SMART_FAST_EARLY_EXIT
        if (excess_left(ex))
#endif
        {
            fill(l1, ex, z);
            fill(l2, ex, z);
            fill(l3, ex, z);
            fill(l4, ex, z);
            fill(l5, ex, z);
        }
        const double df = pD * ex;
        double inf = ex - df;
        double sh = 0.0, dp = 0.0;
        const double p2 = s1 * s1, p3 = p2 * s1, p4 = p2 * p2, p5 = p4 * s1, p6 = p3 * p3;
        leak(l0, s1, s1, p6, inf, sh, dp);
        leak(l1, p2, s1 * 0.5, p5, inf, sh, dp);
        leak(l2, p3, s1 * (1.0 / 3.0), p4, inf, sh, dp);
        leak(l3, p4, s1 * 0.25, p3, inf, sh, dp);
        leak(l4, p5, s1 * 0.2, p2, inf, sh, dp);
        leak(l5, p6, s1 * (1.0 / 6.0), s1, inf, sh, dp);
        if (MERGE) {
            const double xg = sh + dp;
            u_ove = fma(u_ove, dec_s, of + df);
            u_int = fma(u_int, dec_f, inf);
            u_sgw = fma(u_sgw, dec_g, xg);
            if (kBalanceSums)
                xg_sum += xg;
        } else {
            u_ove = clamp(fma(u_ove, dec_s, of * cq_s));
            u_dra = clamp(fma(u_dra, dec_s, df * cq_s));
            u_int = clamp(fma(u_int, dec_f, inf * cq_f));
            u_sgw = clamp(fma(u_sgw, dec_g, sh * cq_g));
            u_dgw = clamp(fma(u_dgw, dec_g, dp * cq_g));
        }
    }

    // ---- wet step of the merged regular variant, leaks by mass balance -------------------------------------------
    // The merged variant needs the interflow leak and the SUM of the two groundwater leaks only.  Each layer still
    // takes its three leaks in the reference's order (l -= l s^i; l -= l s/i; l -= l s^(7-i), structure.py:381-399),
    // one FMA each, but the amounts are not accumulated leak by leak: with F = sum of the layers after filling,
    // A = their sum after the first pass and B = their sum after the third,
    //     interflow leak = F - A,      shallow + deep leak = A - B,
    // and B is the `tot` the next step starts from (:350), F = tot + what infiltrated.  18 + 10 + 3 instructions
    // instead of 36 + 1 + 5.  A difference of layer sums carries their rounding: an absolute ~1e-14 mm per step at
    // levels of ~1e2 mm, <= 1e-11 relative on leaks >= 1e-3 mm -- and the same amount every step while the soil sits
    // at capacity, so a reservoir of constant k collects up to k/dt of them (measured: 1.7e-11 relative on the
    // groundwater total after 7,680 steps).  Fine for a total that is millimetres; not for the deep part on its own,
    // which SPLIT therefore sums directly (deep_leak).  Filling as t = l + ex; l = min(t, z); ex = t - l (3 instead
    // of 4 per layer).
    static constexpr bool kLeakBalance = SMART_FAST_LEAK_BALANCE && MERGE && !GUARD;
    static_assert(kLeakBalance || !SPLIT, "SPLIT is implemented in wet_balance()");

    // Wave-uniform early exits in the filling cascade of a wet step (no lane has excess left after the top layer: 32 %
    // of the wet steps) and in the evaporation cascade of a dry interval.  They trade ~5 vector instructions per step
    // for a compare and a branch: a win when the SIMD has three waves to issue from (vector-ALU bound: -7 % at 1e6
    // samples), a loss when it has one or two (latency bound: +5 % at 1e5).  The launch picks (KArgs::exits).
    static constexpr bool kExits = EXITS;

    // what the third leak pass is about to take from the six layers (structure.py:393-399: l s^(7-i))
    __device__ __forceinline__ double deep_leak(double s1, double p2, double p3, double p4, double p5, double p6) const
    {
        return fma(l5, s1, fma(l4, p2, fma(l3, p3, fma(l2, p4, fma(l1, p5, l0 * p6)))));
    }

    __device__ __forceinline__ static double opaque_zero()
    {
        double x;
        asm volatile("v_mov_b64 %0, 0" : "=v"(x)); // volatile: one of its own for each caller
        return x;
    }

    __device__ static void fill3(double &l, double &ex, double z)
    {
        const double t = l + ex;
        l = fmin(t, z);
        ex = t - l;
    }

    // `tot` = layer sum at the start of the step in, at its end out (same register: see fma_in_place).
    // e_h = ex H / Z is constant over a wet interval.
    __device__ __forceinline__ void wet_balance(const double ex, const double e_h, double &tot)
    {
        const double s1 = sz * tot;
        const double ex_in = fma(-e_h, tot, ex); // excess left after the overland share H tot/Z ex (:363-365)
        double rem = ex_in;
        fill3(l0, rem, z);
        if (!kExits || excess_left(rem)) {
            fill3(l1, rem, z);
            if (!kExits || SMART_FILL_EXITS < 2 || excess_left(rem)) {
                fill3(l2, rem, z);
                if (!kExits || SMART_FILL_EXITS < 3 || excess_left(rem)) {
                    fill3(l3, rem, z);
                    if (!kExits || SMART_FILL_EXITS < 4 || excess_left(rem)) {
                        fill3(l4, rem, z);
                        fill3(l5, rem, z);
                    }
                }
            }
        }
        const double p2 = s1 * s1, p3 = p2 * s1, p4 = p2 * p2, p5 = p4 * s1, p6 = p3 * p3;
        l0 = fma(-l0, s1, l0);
        l1 = fma(-l1, p2, l1);
        l2 = fma(-l2, p3, l2);
        l3 = fma(-l3, p4, l3);
        l4 = fma(-l4, p5, l4);
        l5 = fma(-l5, p6, l5);
        const double after_int = layer_sum();
        l0 = fma(-l0, s1, l0);
        l1 = fma(-l1, s1 * 0.5, l1);
        l2 = fma(-l2, s1 * (1.0 / 3.0), l2);
        l3 = fma(-l3, s1 * 0.25, l3);
        l4 = fma(-l4, s1 * 0.2, l4);
        l5 = fma(-l5, s1 * (1.0 / 6.0), l5);
        // SPLIT: the deep leak is what the third pass takes, summed layer by layer like the reference does (the deep
        // reservoir can be a thousand times smaller than the layers: a difference of layer sums would leave it their
        // rounding, the same amount every step while the soil sits at capacity)
        const double deep = SPLIT ? deep_leak(s1, p2, p3, p4, p5, p6) : 0.0;
        l0 = fma(-l0, p6, l0);
        l1 = fma(-l1, p5, l1);
        l2 = fma(-l2, p4, l2);
        l3 = fma(-l3, p3, l3);
        l4 = fma(-l4, p2, l4);
        l5 = fma(-l5, s1, l5);
        // what entered the layers is ex_in - rem, so  F - A = (tot - A) + ex_in - rem, and with the (1 - D) share of
        // the saturation excess `rem` (:376-377) the interflow reservoir receives (tot - A) + ex_in - D rem
        const double inf = (tot - after_int) + fma(-pD, rem, ex_in);
        fma_in_place(u_ove, dec_s, fma(pD, rem, e_h * tot)); // overland H tot/Z ex + drain D rem
        fma_in_place(u_int, dec_f, inf);
        const double upper = (l0 + l1) + (l2 + l3), lower = l4 + l5;
        asm("v_add_f64 %0, %1, %2" : "+v"(tot) : "v"(upper), "v"(lower)); // tot = layer sum after the step, in place
        const double xg = after_int - tot;
        fma_in_place(u_sgw, dec_g, xg);
        if (SPLIT) {
            fma_in_place(u_dra, dec_s, pD * rem);
            fma_in_place(u_dgw, dec_g, deep);
        }
        if (kBalanceSums)
            xg_sum += xg;
    }

    // the straight-line kernel of low loads (smart_fast_intervals, smart_fast_runs) runs its wet intervals as an asm loop
#ifndef SMART_WET_ASM
#define SMART_WET_ASM 1
#endif
    static constexpr bool kWetAsm = SMART_WET_ASM && kLeakBalance && !kExits && !SPLIT && SMART_FAST_BALANCE_SUMS && !STIFF;

    // `n` wet steps with ZERO excess for every lane (rain == 0 and peva == 0 over the run: wave-uniform, decided by the
    // caller on the scalar unit; never when a layer may be above its capacity): no overland flow, nothing to fill
    __device__ __forceinline__ void calm_interval(long n, double &acc)
    {
        static_assert(kWetAsm, "calm_interval() is the asm loop of the straight-line merged kernels");
        double tot = layer_sum();
        double t0, t1, xf, xg, w_s1, w_p2, w_p3, w_p4, w_p5, w_p6, w_ai;
        int cnt;
        asm volatile(SMART_A_CALM_INTERVAL
                     : [l0] "+v"(l0), [l1] "+v"(l1), [l2] "+v"(l2), [l3] "+v"(l3), [l4] "+v"(l4), [l5] "+v"(l5),
                       [ys] "+v"(u_ove), [yf] "+v"(u_int), [yg] "+v"(u_sgw), [riv] "+v"(u_riv), [acc] "+v"(acc),
                       [tot] "+v"(tot), [xgs] "+v"(xg_sum), [t0] "=&v"(t0), [t1] "=&v"(t1), [xf] "=&v"(xf),
                       [xg] "=&v"(xg), [s1] "=&v"(w_s1), [p2] "=&v"(w_p2), [p3] "=&v"(w_p3), [p4] "=&v"(w_p4),
                       [p5] "=&v"(w_p5), [p6] "=&v"(w_p6), [ai] "=&v"(w_ai), [cnt] "=&s"(cnt)
                     : [cs] "v"(car_s), [cf] "v"(car_f), [cg] "v"(car_g), [oma] "v"(om_ar), [ds] "v"(dec_s),
                       [df] "v"(dec_f), [dg] "v"(dec_g), [sz] "v"(sz), [k3] "s"(-(1.0 / 3.0)), [k5] "s"(-0.2),
                       [k6] "s"(-(1.0 / 6.0)), [n] "s"((int)n)
                     : "scc");
    }

    // `n` wet steps with the same rain excess: the layer sum is handed from step to step
    // `fits`: no layer of this wavefront is above its capacity (wave-uniform; SMART_WET_MODES in smart_fast_arms.h)
    __device__ __forceinline__ void wet_interval(double ex, long n, double &acc, double &num, double &den,
                                                 const bool fits = false)
    {
        if constexpr (kWetAsm) {
            // the whole interval as one asm loop (smart_fast_arms.h: SMART_A_WET_INTERVAL), the arithmetic of
            // route_and_sum() + wet_balance() below, operation for operation
            const double e_h = ex * hz;
            double tot = layer_sum();
            double t0, t1, xs, xf, xg, w_s1, w_p2, w_p3, w_p4, w_p5, w_p6, w_ai;
            int cnt;
#if SMART_WET_E32
            // (the powers carry the sign here -- n_i = -s'^i -- so the second pass's factors are +1/2 ... +1/6 of -s', and
            // -D sits in a register: smart_fast_arms.h, SMART_WET_E32)
            double w_q1, w_q2;
            asm volatile(SMART_A_WET_INTERVAL
                         : [l0] "+v"(l0), [l1] "+v"(l1), [l2] "+v"(l2), [l3] "+v"(l3), [l4] "+v"(l4), [l5] "+v"(l5),
                           [ys] "+v"(u_ove), [yf] "+v"(u_int), [yg] "+v"(u_sgw), [riv] "+v"(u_riv), [acc] "+v"(acc),
                           [tot] "+v"(tot), [xgs] "+v"(xg_sum), [t0] "=&v"(t0), [t1] "=&v"(t1), [xs] "=&v"(xs),
                           [xf] "=&v"(xf), [xg] "=&v"(xg), [s1] "=&v"(w_s1), [p2] "=&v"(w_p2), [p3] "=&v"(w_p3),
                           [p4] "=&v"(w_p4), [p5] "=&v"(w_p5), [p6] "=&v"(w_p6), [ai] "=&v"(w_ai), [q1] "=&v"(w_q1),
                           [q2] "=&v"(w_q2), [cnt] "=&s"(cnt)
                         : [cs] "v"(car_s), [cf] "v"(car_f), [cg] "v"(car_g), [oma] "v"(om_ar), [ds] "v"(dec_s),
                           [df] "v"(dec_f), [dg] "v"(dec_g), [sz] "v"(sz), [z] "v"(z), [pd] "v"(pD), [npd] "v"(-pD),
                           [ex] "v"(ex), [eh] "v"(e_h), [k3] "s"(1.0 / 3.0), [k5] "s"(0.2), [k6] "s"(1.0 / 6.0),
                           [n] "s"((int)n), [ok] "s"(__builtin_amdgcn_readfirstlane((int)fits))
                         : "scc", "vcc");
#else
            asm volatile(SMART_A_WET_INTERVAL
                         : [l0] "+v"(l0), [l1] "+v"(l1), [l2] "+v"(l2), [l3] "+v"(l3), [l4] "+v"(l4), [l5] "+v"(l5),
                           [ys] "+v"(u_ove), [yf] "+v"(u_int), [yg] "+v"(u_sgw), [riv] "+v"(u_riv), [acc] "+v"(acc),
                           [tot] "+v"(tot), [xgs] "+v"(xg_sum), [t0] "=&v"(t0), [t1] "=&v"(t1), [xs] "=&v"(xs),
                           [xf] "=&v"(xf), [xg] "=&v"(xg), [s1] "=&v"(w_s1), [p2] "=&v"(w_p2), [p3] "=&v"(w_p3),
                           [p4] "=&v"(w_p4), [p5] "=&v"(w_p5), [p6] "=&v"(w_p6), [ai] "=&v"(w_ai), [cnt] "=&s"(cnt)
                         : [cs] "v"(car_s), [cf] "v"(car_f), [cg] "v"(car_g), [oma] "v"(om_ar), [ds] "v"(dec_s),
                           [df] "v"(dec_f), [dg] "v"(dec_g), [sz] "v"(sz), [z] "v"(z), [pd] "v"(pD), [ex] "v"(ex),
                           [eh] "v"(e_h), [k3] "s"(-(1.0 / 3.0)), [k5] "s"(-0.2), [k6] "s"(-(1.0 / 6.0)),
                           [n] "s"((int)n), [ok] "s"(__builtin_amdgcn_readfirstlane((int)fits))
                         : "scc", "vcc");
#endif
        } else if (kLeakBalance) {
            const double e_h = ex * hz;
            double tot = layer_sum();
            // unrolled: a taken branch costs a wavefront that has its SIMD (nearly) to itself about as much as a
            // dozen vector instructions (profiles/r01_microbench_valu_salu_branch.txt), and the loop's back-edge
            // is one per 74
#pragma unroll SMART_WET_UNROLL
            for (long k = 0; k < n; ++k) {
                route_and_sum(acc, num, den);
                wet_balance(ex, e_h, tot);
            }
        } else {
            for (long k = 0; k < n; ++k) {
                route_and_sum(acc, num, den);
                wet_lanes(ex);
            }
        }
    }

    // dry branch of structure.py:400-419 for the lanes that are active
    __device__ __forceinline__ void dry_lanes(double ex)
    {
        double d = -ex;
        dry(l0, d, pC);
        // the cascade is over for a lane once nothing is handed down.  With C >= 0 that is "no demand left"; a
        // negative C (GUARD) hands a NEGATIVE demand down, which the next layer takes as an addition
        // (structure.py:410-413: `if lvl >= deficit` holds for any negative deficit), so there only an exact zero ends it
        auto handed_down = [](double x) { return GUARD ? x != 0.0 : x > 0.0; };
#if SMART_FAST_EARLY_EXIT
        // the demand is met by the top layer for every lane of the wave on 57 % of the dry steps, by the top two on
        // 77 %, and reaches the bottom on 11 % (64 random LHS rows, synthetic hourly forcing): two exits, then all
        if (__builtin_amdgcn_ballot_w64(handed_down(d)) != 0)
#endif
        {
            dry(l1, d, pC);
#if SMART_FAST_EARLY_EXIT && SMART_FAST_DRY_EXIT2
            if (__builtin_amdgcn_ballot_w64(handed_down(d)) != 0)
#endif
            {
                dry(l2, d, pC);
                dry(l3, d, pC);
                dry(l4, d, pC);
                dry(l5, d, pC);
            }
        }
        u_ove = clamp(u_ove * dec_s);
        u_int = clamp(u_int * dec_f);
        u_sgw = clamp(u_sgw * dec_g);
        if (!MERGE || SPLIT) {
            u_dra = clamp(u_dra * dec_s);
            u_dgw = clamp(u_dgw * dec_g);
        }
    }

    // River reservoir, outflow sums and the three running sums of the caller.  Called from inside BOTH branches of
    // the step: these ~10 instructions are independent of the soil layers, and placed in the same basic block as
    // the serial fill / evaporation chains they fill the issue slots those chains leave empty.
    __device__ __forceinline__ void route_and_sum(double &acc, double &num, double &den)
    {
        // outflows of this step are the reservoir states at its start (structure.py:427, :487)
        double q_r = u_riv;
        double u_new;
        if (MERGE && !STIFF && kBalanceSums) {
            // the common case spelled out so that the river's old value is used up before it is updated in place (the
            // general form below keeps it in a second register and pays a v_mov_b64 per step for it)
            q_out = u_riv;
            q_gw = cq_g * u_sgw; // (these two: raw reports only -- dead code in the kernels of summary reports)
            q_in = fma(cq_s, u_ove, fma(cq_f, u_int, q_gw));
            const double y = fma(car_s, u_ove, fma(car_f, u_int, car_g * u_sgw));
            if (kExits) { // the kernels with branches in the step (exits; the step loop) need the two welded together
                sum_then_fma_in_place(acc, u_riv, om_ar, y); // (-1.4 %, -2.3 %); the straight-line interval kernel is
            } else {                                         // better off scheduling them itself (welded: +0.9 %)
                acc += u_riv;
                fma_in_place(u_riv, om_ar, y);
            }
            return;
        }
        if (MERGE) {
            // river (structure.py:487-498) in outflow units: U' = U (1 - dt/rk) + dt/rk * sum_j y_j area/1e3/k_j
            u_new = u_riv;
            fma_in_place(u_new, om_ar, fma(car_s, u_ove, fma(car_f, u_int, car_g * u_sgw)));
            q_gw = cq_g * u_sgw; // only the raw-report and non-balance paths read these two
            q_in = fma(cq_s, u_ove, fma(cq_f, u_int, q_gw));
        } else {
            q_gw = u_sgw + u_dgw;
            q_in = ((u_ove + u_dra) + u_int) + q_gw;
            // tmp / rk = U + (q_in - U) * dt / rk
            u_new = fma(q_in - u_riv, a_r, u_riv);
        }
        if (STIFF) {
            if (u_new < 0.0) { // 95 % rule, reachable only when rk < dt
                q_r = 0.95 * fma(u_riv, inv_a_r, q_in);
                u_new = fma(q_in - q_r, a_r, u_riv);
            }
        }
        u_riv = u_new;
        q_out = q_r;
        acc += q_r;
        if (!kBalanceSums) {
            num += q_gw;
            den += q_in;
        }
    }

    // hand-over between two slices of a time-sliced launch (merged variant: 10 states + 3 balance terms)
    static constexpr int kStateFields = 15;

    __device__ void save_state(double *p, int stride) const
    {
        const double v[kStateFields] = {l0, l1, l2, l3, l4, l5, u_ove, u_int, u_sgw, u_riv, g0, r0, xg_sum, u_dra, u_dgw};
#pragma unroll
        for (int i = 0; i < (SPLIT ? 15 : 13); ++i)
            p[i * stride] = v[i];
    }

    __device__ void load_state(const double *p, int stride)
    {
        double v[kStateFields] = {};
#pragma unroll
        for (int i = 0; i < (SPLIT ? 15 : 13); ++i)
            v[i] = p[i * stride];
        l0 = v[0], l1 = v[1], l2 = v[2], l3 = v[3], l4 = v[4], l5 = v[5];
        u_ove = v[6], u_int = v[7], u_sgw = v[8], u_riv = v[9];
        g0 = v[10], r0 = v[11], xg_sum = v[12];
        u_dra = v[13], u_dgw = v[14];
        note_capacity();
    }

    // ---- groundwater ratio without per-step sums (regular merged variant) -------------------------------------
    // Both sums of structure.py:191 follow from the linear updates themselves:
    //   river  U' = U + (q_in - U) a_r        =>  sum_t q_in = sum_t U(t) + (U(T) - U(0)) / a_r,  sum_t U(t) = sum of Q_out
    //   gw     G' = G (1 - a_g) + x cq_g      =>  sum_t G(t) = (G(0) - G(T) + cq_g sum_t x(t)) / a_g
    // so the step only adds the groundwater inflow on wet steps; num / den are assembled once at the end.
    static constexpr bool kBalanceSums = SMART_FAST_BALANCE_SUMS && MERGE;
    double g0, r0, xg_sum;

    __device__ void begin_run()
    {
        g0 = u_sgw;
        r0 = u_riv;
        xg_sum = 0.0;
    }

    __device__ void balance_sums(double q_out_total, double &num, double &den) const
    {
        num = cq_g * (xg_sum + (g0 - u_sgw)) / (1.0 - dec_g); // the reservoir is carried in mm: outflow = y cq_g
        den = fma(u_riv - r0, inv_a_r, q_out_total);
    }

    __device__ void step(double /*rain_in*/, double /*peva_in*/, double ex, double &acc, double &num, double &den)
    {
        if (ex >= 0.0) { // structure.py:359
            route_and_sum(acc, num, den);
            wet_lanes(ex);
        } else { // :400
            route_and_sum(acc, num, den);
            dry_lanes(ex);
        }
    }

    __device__ void step(double r, double e, double ex, double &acc, double &num, double &den, StepGeneric)
    {
        step(r, e, ex, acc, num, den);
    }

    // every lane of the wavefront is wet for the whole report interval: no compare, no branch
    __device__ void step(double, double, double ex, double &acc, double &num, double &den, StepAllWet)
    {
        route_and_sum(acc, num, den);
        wet_lanes(ex);
    }

    // ---- step loop with deferred evaporation (forcing that varies inside the report interval) -----------------------
    // (step_lazy() below is the compiled form of round 2, kept behind -DSMART_STEP_ARMS=0 as the bit-for-bit yardstick
    // of the asm arms further down, which perform the same operations in the same order)
    // A dry step does two unrelated things: the reservoirs drain (routing, needed every step: the river's outflow is
    // reported) and the evaporation demand is taken from the soil layers (6 x 3 dependent instructions) -- but nothing
    // looks at the layers again until the lane's next WET step.  The cascade composes additively (dry_interval below:
    // demands d1 then d2 leave the same levels and hand down the same total as d1 + d2 at once), so a dry step only
    // adds its demand to `pend` and the cascade runs once, when the lane turns wet (or its states are asked for).
    // Lanes with nothing pending see the cascade as the identity, so a sample's arithmetic does not depend on its
    // wave neighbours.  The layer sum is carried from wet step to wet step (`tot_c`, recomputed after a cascade).
    // Steps with neither rain nor evaporation for the whole wavefront (night hours of sub-daily data; forcing is
    // wave-uniform) are wet steps with zero excess: no overland flow, nothing to fill, only the leaks -- the filling
    // cascade is skipped, bit-identical to running it with zero excess as long as no layer is above its capacity (a
    // caller's initial state could be: zero_ok).
    double pend, tot_c;
    bool zero_ok;
    // lanes that may have something pending (SGPR pair): which side of zero a step's excess falls on is known as a
    // lane mask a chunk of steps ahead (time_loop_masks), so whether the cascade has to run is decided on the scalar
    // unit from masks that have long been there -- not by comparing `pend` and waiting for the answer
    unsigned long long pend_mask;

    __device__ void begin_lazy(double pending)
    {
        pend = pending;
        pend_mask = __builtin_amdgcn_ballot_w64(pending > 0.0);
        tot_c = layer_sum();
        note_capacity();
        zero_ok = over_mask == 0;
    }

    __device__ __forceinline__ void flush_pending()
    {
        // the exits only skip identity operations (a lane whose demand is met sees max(l, 0) and max(-C l, 0) = 0)
        double d = pend;
        dry(l0, d, pC);
        if (__builtin_amdgcn_ballot_w64(d > 0.0) != 0) {
            dry(l1, d, pC);
            if (__builtin_amdgcn_ballot_w64(d > 0.0) != 0) {
                dry(l2, d, pC);
                dry(l3, d, pC);
                dry(l4, d, pC);
                dry(l5, d, pC);
            }
        }
        pend = 0.0;
        tot_c = layer_sum();
    }

    // One step, written as straight-line code around ONE predicated region: every lane routes, every lane adds its
    // (possibly zero) evaporation demand to `pend` and its (possibly zero) inflows to the reservoirs; only the soil
    // half of a wet step sits under `if (ex >= 0)`.  (A three-way wet / calm / dry branch with the routing in each arm
    // cost more than it saved: the arms left the states in different registers and the joins paid for it.)
    // `calm`: rain == 0 and peva == 0 for this step and zero_ok (wave-uniform, decided on the scalar unit by the
    // caller).
    // `wet`: ballot of ex >= 0 over the wavefront, worked out ahead of the step by the caller
    __device__ __forceinline__ void step_lazy(double ex, unsigned long long wet, bool calm, double &acc, double &num,
                                              double &den)
    {
        route_and_sum(acc, num, den);
        pend += fmax(-ex, 0.0);
        // (the zeros of x_s and x_f as values the compiler cannot see through: a calm step then finds them where the
        // dry lanes' zeros are, instead of in a block of its own that sets them again, and the filling code falls
        // through into the leaks)
        double x_s = opaque_zero(), x_f = opaque_zero(), x_g = 0.0, x_dra = 0.0, x_dgw = 0.0;
        // after a step the lanes with something pending are exactly the ones that were on its dry side: the cascade
        // has to run when a lane that was dry on the previous step is wet on this one
        const bool cascade = (pend_mask & wet) != 0;
        pend_mask = ~wet;
        if (__builtin_amdgcn_inverse_ballot_w64(wet)) { // structure.py:359
            if (__builtin_expect(cascade, 0))
                flush_pending();
            double &tot = tot_c;
            const double s1 = sz * tot;
            if (__builtin_expect(!calm, 1)) { // scalar branch: nothing to fill on a calm step (zero_ok is in `calm`)
                const double e_h = ex * hz;
                const double ex_in = fma(-e_h, tot, ex); // excess left after the overland share H tot/Z ex (:363-365)
                double rem = ex_in;
                fill3(l0, rem, z);
                fill3(l1, rem, z);
                fill3(l2, rem, z);
                fill3(l3, rem, z);
                fill3(l4, rem, z);
                fill3(l5, rem, z);
                // what the quick and the inter reservoir get from the rain itself; a calm step leaves the zeros the
                // dry lanes use (no block of its own to set them: the fill falls through into the leaks)
                x_f = fma(-pD, rem, ex_in);
                x_s = fma(pD, rem, e_h * tot);
                if (SPLIT)
                    x_dra = pD * rem;
            }
            const double p2 = s1 * s1, p3 = p2 * s1, p4 = p2 * p2, p5 = p4 * s1, p6 = p3 * p3;
            l0 = fma(-l0, s1, l0);
            l1 = fma(-l1, p2, l1);
            l2 = fma(-l2, p3, l2);
            l3 = fma(-l3, p4, l3);
            l4 = fma(-l4, p5, l4);
            l5 = fma(-l5, p6, l5);
            const double after_int = layer_sum();
            l0 = fma(-l0, s1, l0);
            l1 = fma(-l1, s1 * 0.5, l1);
            l2 = fma(-l2, s1 * (1.0 / 3.0), l2);
            l3 = fma(-l3, s1 * 0.25, l3);
            l4 = fma(-l4, s1 * 0.2, l4);
            l5 = fma(-l5, s1 * (1.0 / 6.0), l5);
            const double deep = SPLIT ? deep_leak(s1, p2, p3, p4, p5, p6) : 0.0;
            l0 = fma(-l0, p6, l0);
            l1 = fma(-l1, p5, l1);
            l2 = fma(-l2, p4, l2);
            l3 = fma(-l3, p3, l3);
            l4 = fma(-l4, p2, l4);
            l5 = fma(-l5, s1, l5);
            x_f = (tot - after_int) + x_f;
            tot = layer_sum();
            x_g = after_int - tot;
            if (SPLIT)
                x_dgw = deep;
        }
        fma_in_place(u_ove, dec_s, x_s);
        fma_in_place(u_int, dec_f, x_f);
        fma_in_place(u_sgw, dec_g, x_g);
        if (SPLIT) {
            fma_in_place(u_dra, dec_s, x_dra);
            fma_in_place(u_dgw, dec_g, x_dgw);
        }
        xg_sum += x_g;
    }

    // ---- the step loop at the instruction level (smart_fast_arms.h has the text and the story) ----------------------
#define SMART_ARM_STATES                                                                                               \
    [l0] "+v"(l0), [l1] "+v"(l1), [l2] "+v"(l2), [l3] "+v"(l3), [l4] "+v"(l4), [l5] "+v"(l5), [ys] "+v"(u_ove),       \
        [yf] "+v"(u_int), [yg] "+v"(u_sgw), [riv] "+v"(u_riv), [acc] "+v"(acc), [tot] "+v"(tot_c), [pend] "+v"(pend),  \
        [xgs] "+v"(xg_sum)
#define SMART_ARM_TEMPS                                                                                                \
    [t0] "=&v"(t0), [t1] "=&v"(t1), [ex] "=&v"(ex), [eh] "=&v"(eh), [xs] "=&v"(xs), [xf] "=&v"(xf), [xg] "=&v"(xg),   \
        [p4] "=&v"(w_p4), [p5] "=&v"(w_p5), [p6] "=&v"(w_p6), [ai] "=&v"(w_ai), [wm] "=&s"(wm), [sv] "=&s"(sv),       \
        [tmp] "=&s"(tmp)
#define SMART_ARM_SPLIT [yd] "+v"(u_dra), [ydg] "+v"(u_dgw), [xd] "=&v"(xd), [dp] "=&v"(dp)
#define SMART_ARM_CONSTS                                                                                               \
    [cs] "v"(car_s), [cf] "v"(car_f), [cg] "v"(car_g), [oma] "v"(om_ar), [ds] "v"(dec_s), [df] "v"(dec_f),            \
        [dg] "v"(dec_g), [sz] "v"(sz), [hz] "v"(hz), [z] "v"(z), [pd] "v"(pD), [pc] "v"(pC), [pt] "v"(pT),            \
        [k3] "s"(-(1.0 / 3.0)), [k5] "s"(-0.2), [k6] "s"(-(1.0 / 6.0))
#define SMART_ARM_LOCALS                                                                                               \
    double t0, t1, ex, eh, xs, xf, xg, w_p4, w_p5, w_p6, w_ai;                                                         \
    unsigned long long wm, sv, tmp

    // One step.  QUICK: the forcing is sane and no layer of this wave is above capacity -- dispatch over the three
    // arms; otherwise the rain arm.  LAST: the routing of SMART_A_ROUTE_LAST -- `acc` = the river's outflow of this step,
    // q_gw / q_in = dt/rk times its groundwater inflow / its whole inflow (raw reports, a report every step).
#define SMART_ARM_LAST [qg] "=&v"(q_gw), [qi] "=&v"(q_in)
    template <bool QUICK, bool LAST = false>
    __device__ __forceinline__ void step_arms(const double2 v, double &acc)
    {
        SMART_ARM_LOCALS;
        if constexpr (LAST) {
            static_assert(!SPLIT, "raw reports / a report every step with the final state vector take smart_fast_plain");
            if constexpr (QUICK)
                asm volatile(SMART_A_STEP(SMART_A_ROUTE_LAST, "", "", "", "", "", "")
                             : SMART_ARM_STATES, SMART_ARM_TEMPS, SMART_ARM_LAST
                             : SMART_ARM_CONSTS, [rn0] "s"(v.x), [pe0] "s"(v.y)
                             : "vcc", "scc");
            else
                asm volatile(SMART_A_STEP_RAIN(SMART_A_ROUTE_LAST, "", "", "", "")
                             : SMART_ARM_STATES, SMART_ARM_TEMPS, SMART_ARM_LAST
                             : SMART_ARM_CONSTS, [rn0] "s"(v.x), [pe0] "s"(v.y)
                             : "vcc", "scc");
        } else if constexpr (SPLIT) {
            double xd, dp;
            if constexpr (QUICK)
                asm volatile(SMART_A_STEP(SMART_A_ROUTE, SMART_A_DEEP, SMART_A_CALM_SPLIT, SMART_A_RAIN_ZEROS_SPLIT,
                                          SMART_A_RAIN_DRAIN_SPLIT, SMART_A_RAIN_SPLIT, SMART_A_DRY_SPLIT)
                             : SMART_ARM_STATES, SMART_ARM_TEMPS, SMART_ARM_SPLIT
                             : SMART_ARM_CONSTS, [rn0] "s"(v.x), [pe0] "s"(v.y)
                             : "vcc", "scc");
            else
                asm volatile(SMART_A_STEP_RAIN(SMART_A_ROUTE, SMART_A_DEEP, SMART_A_RAIN_ZEROS_SPLIT,
                                               SMART_A_RAIN_DRAIN_SPLIT, SMART_A_RAIN_SPLIT)
                             : SMART_ARM_STATES, SMART_ARM_TEMPS, SMART_ARM_SPLIT
                             : SMART_ARM_CONSTS, [rn0] "s"(v.x), [pe0] "s"(v.y)
                             : "vcc", "scc");
        } else {
            if constexpr (QUICK)
                asm volatile(SMART_A_STEP(SMART_A_ROUTE, "", "", "", "", "", "")
                             : SMART_ARM_STATES, SMART_ARM_TEMPS
                             : SMART_ARM_CONSTS, [rn0] "s"(v.x), [pe0] "s"(v.y)
                             : "vcc", "scc");
            else
                asm volatile(SMART_A_STEP_RAIN(SMART_A_ROUTE, "", "", "", "")
                             : SMART_ARM_STATES, SMART_ARM_TEMPS
                             : SMART_ARM_CONSTS, [rn0] "s"(v.x), [pe0] "s"(v.y)
                             : "vcc", "scc");
        }
    }

    // The four steps of a chunk, threaded (SMART_A_CHUNK).
#define SMART_ARM_CHUNK_IN                                                                                             \
    [rn0] "s"(c[0].x), [pe0] "s"(c[0].y), [rn1] "s"(c[1].x), [pe1] "s"(c[1].y), [rn2] "s"(c[2].x), [pe2] "s"(c[2].y), \
        [rn3] "s"(c[3].x), [pe3] "s"(c[3].y)
    template <bool QUICK, bool LAST = false>
    __device__ __forceinline__ void chunk_arms(const double2 (&c)[4], double &acc)
    {
        SMART_ARM_LOCALS;
        if constexpr (LAST) {
            static_assert(!SPLIT, "raw reports / a report every step with the final state vector take smart_fast_plain");
            if constexpr (QUICK)
                asm volatile(SMART_A_CHUNK(SMART_A_ROUTE_LAST, "", "", "", "", "", "")
                             : SMART_ARM_STATES, SMART_ARM_TEMPS, SMART_ARM_LAST
                             : SMART_ARM_CONSTS, SMART_ARM_CHUNK_IN
                             : "vcc", "scc");
            else
                asm volatile(SMART_A_CHUNK_RAIN(SMART_A_ROUTE_LAST, "", "", "", "")
                             : SMART_ARM_STATES, SMART_ARM_TEMPS, SMART_ARM_LAST
                             : SMART_ARM_CONSTS, SMART_ARM_CHUNK_IN
                             : "vcc", "scc");
        } else if constexpr (SPLIT) {
            double xd, dp;
            if constexpr (QUICK)
                asm volatile(SMART_A_CHUNK(SMART_A_ROUTE, SMART_A_DEEP, SMART_A_CALM_SPLIT, SMART_A_RAIN_ZEROS_SPLIT,
                                           SMART_A_RAIN_DRAIN_SPLIT, SMART_A_RAIN_SPLIT, SMART_A_DRY_SPLIT)
                             : SMART_ARM_STATES, SMART_ARM_TEMPS, SMART_ARM_SPLIT
                             : SMART_ARM_CONSTS, SMART_ARM_CHUNK_IN
                             : "vcc", "scc");
            else
                asm volatile(SMART_A_CHUNK_RAIN(SMART_A_ROUTE, SMART_A_DEEP, SMART_A_RAIN_ZEROS_SPLIT,
                                                SMART_A_RAIN_DRAIN_SPLIT, SMART_A_RAIN_SPLIT)
                             : SMART_ARM_STATES, SMART_ARM_TEMPS, SMART_ARM_SPLIT
                             : SMART_ARM_CONSTS, SMART_ARM_CHUNK_IN
                             : "vcc", "scc");
        } else {
            if constexpr (QUICK)
                asm volatile(SMART_A_CHUNK(SMART_A_ROUTE, "", "", "", "", "", "")
                             : SMART_ARM_STATES, SMART_ARM_TEMPS
                             : SMART_ARM_CONSTS, SMART_ARM_CHUNK_IN
                             : "vcc", "scc");
            else
                asm volatile(SMART_A_CHUNK_RAIN(SMART_A_ROUTE, "", "", "", "")
                             : SMART_ARM_STATES, SMART_ARM_TEMPS
                             : SMART_ARM_CONSTS, SMART_ARM_CHUNK_IN
                             : "vcc", "scc");
        }
    }

    // A stretch of `n_iv` report intervals of `half` pairs of chunks each, walked pair of steps by pair of steps through the
    // blocks of SMART_A_PAIRS_STRETCH, the report at the end of every interval included (Reporter::emit and what
    // run_ensemble_merged's report() keeps, operation for operation): ONE asm, no compiled code until the stretch is over.
    // f_asm / codes: the first interval's first chunk in the forcing (through a pointer that hipcc does not take for the
    // __restrict__ one: smart_device.h) and its code words (smart_forcing_scan); obs_p / dev_p: the first interval's
    // observation and deviation (null: none, or the warm-up -- reporting: false).  first_is_0: the stretch's first report
    // is report 0, which sets the constant the moments are taken about.  sum_a / sum_b: the sum of the outflows (means;
    // sum_b unused) or the two sums of the raw groundwater ratio (LAST).  QUICK waves only; the asm requests two chunks
    // beyond the stretch: the caller keeps the last interval of the forcing array away from it.
    // ODD: intervals of any whole number of chunks (half = that number; the stretch may start in either register buffer:
    // first_chunk_odd); otherwise an even number (half = half of it).
    template <bool LAST, bool ODD = false>
    __device__ __forceinline__ void stream_stretch(const double2 *f_asm, const uint2 *codes, const double *obs_p,
                                                   const double *dev_p, long n_iv, int half, bool first_chunk_odd,
                                                   bool reporting, bool storing,
                                                   bool first_is_0, double inv_gap, double &acc, double &mA, double &mB,
                                                   double &mC1, double &mC2, double &mC3, double &shift, double &sum_a,
                                                   double &sum_b, double *&row, long ld)
    {
        static_assert(!(SPLIT && LAST), "raw reports with the final state vector take smart_fast_plain");
        SMART_ARM_LOCALS;
        const int has_obs = obs_p != nullptr;
        if (!obs_p) {
            obs_p = reinterpret_cast<const double *>(f_asm); // (somewhere to load from: n_iv doubles of the forcing)
            dev_p = obs_p;
        }
        // (wave-uniform, but hipcc does not see it through the caller's closures: a vector register is no base of a load)
        auto uniform = [](const double *p) {
            const unsigned long long u = (unsigned long long)p;
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
            return (const double *)(((unsigned long long)hi << 32) | lo);
        };
        obs_p = uniform(obs_p);
        dev_p = uniform(dev_p);
        double rv, rd, ru;
        unsigned long long row_bits = (unsigned long long)row;
        // (wave-uniform all of them; readfirstlane tells hipcc)
        const int n = __builtin_amdgcn_readfirstlane((int)n_iv), rep_i = __builtin_amdgcn_readfirstlane((int)reporting),
                  sto_i = __builtin_amdgcn_readfirstlane((int)(storing && reporting)),
                  hob_i = __builtin_amdgcn_readfirstlane((int)(has_obs && reporting)),
                  r0_i = __builtin_amdgcn_readfirstlane((int)(first_is_0 && reporting)),
                  par_i = __builtin_amdgcn_readfirstlane((int)(ODD && first_chunk_odd));
#define SMART_STRETCH_OUT                                                                                              \
    [mA] "+v"(mA), [mB] "+v"(mB), [mC1] "+v"(mC1), [mC2] "+v"(mC2), [mC3] "+v"(mC3), [shift] "+v"(shift),              \
        [row] "+v"(row_bits), [rv] "=&v"(rv), [rd] "=&v"(rd), [ru] "=&v"(ru)
#define SMART_STRETCH_IN                                                                                               \
    [fp] "s"(f_asm), [cp] "s"(codes), [half] "s"(half), [niv] "s"(n), [op] "s"(obs_p), [wp] "s"(dev_p),                \
        [rep] "s"(rep_i), [sto] "s"(sto_i), [hob] "s"(hob_i), [r0] "s"(r0_i), [par] "s"(par_i), [ig] "v"(inv_gap),     \
        [ld] "s"(ld)
#define SMART_STRETCH_ASM(STRETCH)                                                                                     \
    if constexpr (LAST)                                                                                                \
        asm volatile(STRETCH(SMART_A_ROUTE_LAST, SMART_P_REPORT_LAST, "", "", "", "", "", "")                          \
                     : SMART_ARM_STATES, SMART_ARM_TEMPS, SMART_ARM_LAST, SMART_STRETCH_OUT, [numr] "+v"(sum_a),       \
                       [denr] "+v"(sum_b)                                                                              \
                     : SMART_ARM_CONSTS, SMART_STRETCH_IN                                                              \
                     : SMART_S_CLOBBERS);                                                                              \
    else                                                                                                               \
        asm volatile(STRETCH(SMART_A_ROUTE, SMART_P_REPORT_MEAN, "", "", "", "", "", "")                               \
                     : SMART_ARM_STATES, SMART_ARM_TEMPS, SMART_STRETCH_OUT, [qtot] "+v"(sum_a)                        \
                     : SMART_ARM_CONSTS, SMART_STRETCH_IN                                                              \
                     : SMART_S_CLOBBERS)
        if constexpr (SPLIT) {
            // the models with the final state vector: two more reservoirs in every arm, blocks SMART_PS_STRIDE bytes apart
            // (the macros read SMART_P_STRIDE where they are expanded: here)
            double xd, dp;
#pragma push_macro("SMART_P_STRIDE")
#undef SMART_P_STRIDE
#define SMART_P_STRIDE SMART_PS_STRIDE
            if constexpr (ODD)
                asm volatile(SMART_A_PAIRS_STRETCH_ODD(SMART_A_ROUTE, SMART_P_REPORT_MEAN, SMART_A_DEEP, SMART_A_CALM_SPLIT,
                                                       SMART_A_RAIN_ZEROS_SPLIT, SMART_A_RAIN_DRAIN_SPLIT,
                                                       SMART_A_RAIN_SPLIT, SMART_A_DRY_SPLIT)
                             : SMART_ARM_STATES, SMART_ARM_TEMPS, SMART_ARM_SPLIT, SMART_STRETCH_OUT, [qtot] "+v"(sum_a)
                             : SMART_ARM_CONSTS, SMART_STRETCH_IN
                             : SMART_S_CLOBBERS);
            else
                asm volatile(SMART_A_PAIRS_STRETCH(SMART_A_ROUTE, SMART_P_REPORT_MEAN, SMART_A_DEEP, SMART_A_CALM_SPLIT,
                                                   SMART_A_RAIN_ZEROS_SPLIT, SMART_A_RAIN_DRAIN_SPLIT, SMART_A_RAIN_SPLIT,
                                                   SMART_A_DRY_SPLIT)
                             : SMART_ARM_STATES, SMART_ARM_TEMPS, SMART_ARM_SPLIT, SMART_STRETCH_OUT, [qtot] "+v"(sum_a)
                             : SMART_ARM_CONSTS, SMART_STRETCH_IN
                             : SMART_S_CLOBBERS);
#pragma pop_macro("SMART_P_STRIDE")
        } else if constexpr (ODD) {
            SMART_STRETCH_ASM(SMART_A_PAIRS_STRETCH_ODD);
        } else {
            SMART_STRETCH_ASM(SMART_A_PAIRS_STRETCH);
        }
        row = (double *)row_bits;
    }

    // `quads` x 4 steps with a report after every step, walked through the blocks of SMART_A_EVERY_STREAM: records = the
    // stretch's first pair in smart_forcing_scan's stream (rain, PE, observation, deviation per step), codes = its code
    // word.  The report's variables are Reporter's (smart_device.h: report_every); row = this lane's place in the first
    // report's row of the discharge matrix, moved on by ld per report.  QUICK waves, models without the final row.
    template <bool STORE, bool OBS>
    __device__ __forceinline__ void stream_every(const double *records, const unsigned *codes, int quads, double &acc,
                                                 double &mA, double &mB, double &mC1, double &mC2, double &mC3,
                                                 const double shift, double &qtot, double *&row, const long ld)
    {
        static_assert(!SPLIT, "stream_every: the SPLIT models take time_loop_arms_each");
        SMART_ARM_LOCALS;
        double rd, ru;
        unsigned long long row_bits = (unsigned long long)row;
#define SMART_EVERY_OUT                                                                                                \
    SMART_ARM_STATES, SMART_ARM_TEMPS, SMART_ARM_LAST, [mA] "+v"(mA), [mB] "+v"(mB), [mC1] "+v"(mC1), [mC2] "+v"(mC2),   \
        [mC3] "+v"(mC3), [qtot] "+v"(qtot), [row] "+v"(row_bits), [rd] "=&v"(rd), [ru] "=&v"(ru)
#define SMART_EVERY_IN                                                                                                 \
    SMART_ARM_CONSTS, [sp] "s"(records), [cp] "s"(codes), [quads] "s"(quads), [shift] "v"(shift), [ld] "s"(ld)
        if constexpr (STORE && OBS)
            asm volatile(SMART_A_EVERY_STREAM(SMART_E_STORE, SMART_E_MOMENTS) : SMART_EVERY_OUT : SMART_EVERY_IN : SMART_E_CLOBBERS);
        else if constexpr (STORE)
            asm volatile(SMART_A_EVERY_STREAM(SMART_E_STORE, SMART_E_NOMOM) : SMART_EVERY_OUT : SMART_EVERY_IN : SMART_E_CLOBBERS);
        else if constexpr (OBS)
            asm volatile(SMART_A_EVERY_STREAM("", SMART_E_MOMENTS) : SMART_EVERY_OUT : SMART_EVERY_IN : SMART_E_CLOBBERS);
        else
            asm volatile(SMART_A_EVERY_STREAM("", SMART_E_NOMOM) : SMART_EVERY_OUT : SMART_EVERY_IN : SMART_E_CLOBBERS);
        row = (double *)row_bits;
    }

    // `quads` x 4 steps of a run whose report gap is not a whole number of chunks, through the blocks of SMART_A_GAP_STREAM:
    // the stream of records and code words of stream_every(), the interval's report (stream_stretch()'s: flags, report 0's
    // constant, row pointer, moments, sums) behind the arms whose steps end an interval -- the code words know which.
    // The steps are whole intervals (the caller sees to it); acc, the interval's sum, is zero going in and coming out.
    template <bool LAST>
    __device__ __forceinline__ void stream_gap(const double *records, const unsigned *codes, int quads, bool reporting,
                                               bool storing, bool has_obs, bool first_is_0, double inv_gap, double &acc,
                                               double &mA, double &mB, double &mC1, double &mC2, double &mC3, double &shift,
                                               double &sum_a, double &sum_b, double *&row, long ld)
    {
        static_assert(!SPLIT, "stream_gap: the SPLIT models take the threaded chunks");
        SMART_ARM_LOCALS;
        double rv, rd, ru;
        unsigned long long row_bits = (unsigned long long)row;
        const int q = __builtin_amdgcn_readfirstlane(quads), rep_i = __builtin_amdgcn_readfirstlane((int)reporting),
                  sto_i = __builtin_amdgcn_readfirstlane((int)(storing && reporting)),
                  hob_i = __builtin_amdgcn_readfirstlane((int)(has_obs && reporting)),
                  r0_i = __builtin_amdgcn_readfirstlane((int)(first_is_0 && reporting));
#define SMART_GAP_IN                                                                                                   \
    SMART_ARM_CONSTS, [sp] "s"(records), [cp] "s"(codes), [quads] "s"(q), [rep] "s"(rep_i), [sto] "s"(sto_i),          \
        [hob] "s"(hob_i), [r0] "s"(r0_i), [ig] "v"(inv_gap), [ld] "s"(ld)
        if constexpr (LAST)
            asm volatile(SMART_A_GAP_STREAM(SMART_A_ROUTE_LAST, SMART_P_VALUE_LAST, SMART_P_AFTER_LAST)
                         : SMART_ARM_STATES, SMART_ARM_TEMPS, SMART_ARM_LAST, SMART_STRETCH_OUT, [numr] "+v"(sum_a),
                           [denr] "+v"(sum_b)
                         : SMART_GAP_IN
                         : SMART_S_CLOBBERS);
        else
            asm volatile(SMART_A_GAP_STREAM(SMART_A_ROUTE, SMART_P_VALUE_MEAN, SMART_P_AFTER_MEAN)
                         : SMART_ARM_STATES, SMART_ARM_TEMPS, SMART_STRETCH_OUT, [qtot] "+v"(sum_a)
                         : SMART_GAP_IN
                         : SMART_S_CLOBBERS);
        row = (double *)row_bits;
    }

    // ---- a whole report interval without rain excess (run_ensemble_merged) ------------------------------------
    // While no lane gets inflow the routing half of the model is linear with constant coefficients:
    //   U_j' = dec_j U_j  (j = quick, inter, groundwater),   U_riv' = (1 - a_r) U_riv + a_r (U_q + U_i + U_g)
    // so n steps are one map  U_j(n) = P_j U_j,  U_riv(n) = P_r U_riv + sum_j A_j U_j,  and the interval's sum of
    // river outflows is  B_r U_riv + sum_j B_j U_j.  The eleven coefficients are built once per sample by running
    // the recurrence itself n times on unit vectors (no closed form: no cancellation when 1 - a_r is close to dec_j).
    static constexpr bool kIntervals = MERGE && kBalanceSums;
    double P_q, P_i, P_g, P_r, A_q, A_i, A_g, B_q, B_i, B_g, B_r;

    __device__ void setup_intervals(long n)
    {
        const double b = 1.0 - a_r;
        double pq = 1.0, pi = 1.0, pg = 1.0, pr = 1.0;
        double aq = 0.0, ai = 0.0, ag = 0.0;
        double bq = 0.0, bi = 0.0, bg = 0.0, br = 0.0;
        for (long k = 0; k < n; ++k) {
            bq += aq; // contribution of U_q(0) to U_riv(k), summed over k
            bi += ai;
            bg += ag;
            br += pr;
            aq = fma(b, aq, a_r * pq);
            ai = fma(b, ai, a_r * pi);
            ag = fma(b, ag, a_r * pg);
            pr *= b;
            pq *= dec_s;
            pi *= dec_f;
            pg *= dec_g;
        }
        P_q = pq, P_i = pi, P_g = pg, P_r = pr;
        // the reservoirs are carried as volumes in mm, their outflows are y * area/1e3/k
        A_q = aq * cq_s, A_i = ai * cq_f, A_g = ag * cq_g;
        B_q = bq * cq_s, B_i = bi * cq_f, B_g = bg * cq_g, B_r = br;
    }

    // `n` consecutive dry steps with the same demand d0 = -ex on every one of them; adds the n river outflows to acc
    __device__ __forceinline__ void dry_interval(double ex, long n, double &acc)
    {
        // The evaporation cascade composes additively: a layer maps a demand d to (max(l - d, 0), C max(d - l, 0)),
        // so demands d1 then d2 leave the same level and hand down the same total as d1 + d2 at once, and by
        // induction over the layers n steps of demand d0 equal one step of demand n d0 (structure.py:409-419;
        // the two differ by the rounding of n subtractions).  Every lane does the same thing, so a sample's
        // arithmetic does not depend on its wave neighbours; the early exits only skip identity operations.
        double d = -ex * (double)n;
#if SMART_IV_DEFER
        // ... and, by the same argument, the demands of CONSECUTIVE dry intervals add up as well: nothing looks at the
        // layers before the lane's next wet interval, so the demand only joins `pend` here and the cascade runs once
        // per dry spell, in front of that wet interval (run_ensemble_merged) -- 13 instead of 36 vector instructions
        // for a dry interval.  `pend` travels in the slice hand-over like the states.
        pend += d;
#else
        dry(l0, d, pC);
        if (!kExits || __builtin_amdgcn_ballot_w64(d > 0.0) != 0) {
            dry(l1, d, pC);
            if (!kExits || __builtin_amdgcn_ballot_w64(d > 0.0) != 0) {
                dry(l2, d, pC);
                dry(l3, d, pC);
                dry(l4, d, pC);
                dry(l5, d, pC);
            }
        }
#endif
        acc += fma(B_r, u_riv, fma(B_q, u_ove, fma(B_i, u_int, B_g * u_sgw)));
        u_riv = fma(P_r, u_riv, fma(A_q, u_ove, fma(A_i, u_int, A_g * u_sgw)));
        u_ove *= P_q;
        u_int *= P_i;
        u_sgw *= P_g;
        if (SPLIT) {
            u_dra *= P_q;
            u_dgw *= P_g;
        }
    }

    // ---- raw reports over piecewise-constant forcing (smart_fast_intervals_raw) -----------------------------------
    // A raw report is the river's outflow of the LAST step of its interval (structure.py:192-195), and the groundwater
    // ratio of a raw run sums the flows of those steps only: the interval engine advances n - 1 steps at once (the
    // coefficients of the dry map are then built for n - 1 steps: setup_intervals(n - 1)), looks at the reservoirs --
    // outflows of a step are the states it starts from (:427, :487) -- and takes the last step on its own.
    // qo: the river's outflow; qi / qg: dt/rk times the river's inflow / its groundwater part (the common factor
    // cancels in the ratio), the same quantities SMART_A_ROUTE_LAST leaves behind in the step loop.
    __device__ __forceinline__ void last_step_flows(double &qo, double &qg, double &qi) const
    {
        qg = car_g * u_sgw;
        qi = fma(car_s, u_ove, fma(car_f, u_int, qg));
        qo = u_riv;
    }

    // `n` consecutive dry steps with the same demand; the routing as the map of n - 1 steps + the last step
    __device__ __forceinline__ void dry_interval_last(double ex, long n, double &qo, double &qg, double &qi)
    {
        static_assert(!SPLIT && MERGE && !STIFF, "the merged regular variant");
        double d = -ex * (double)n; // the evaporation cascade composes additively: dry_interval()
        dry(l0, d, pC);
        if (!kExits || __builtin_amdgcn_ballot_w64(d > 0.0) != 0) {
            dry(l1, d, pC);
            if (!kExits || __builtin_amdgcn_ballot_w64(d > 0.0) != 0) {
                dry(l2, d, pC);
                dry(l3, d, pC);
                dry(l4, d, pC);
                dry(l5, d, pC);
            }
        }
        u_riv = fma(P_r, u_riv, fma(A_q, u_ove, fma(A_i, u_int, A_g * u_sgw)));
        u_ove *= P_q;
        u_int *= P_i;
        u_sgw *= P_g;
        last_step_flows(qo, qg, qi);
        fma_in_place(u_riv, om_ar, qi);
        u_ove *= dec_s;
        u_int *= dec_f;
        u_sgw *= dec_g;
    }
};

// Which instantiation does this wavefront need?  Decided once from its own 64 parameter rows.
__device__ inline int wave_class(const KArgs &a, long block, long catchment)
{
    long n = block * kWave + threadIdx.x;
    if (n >= a.N)
        n = a.N - 1;
    const double *p = a.params + catchment * a.pstride_c + n * 10;
    const double dt = a.dt;
    const bool stiff = !(p[6] * 3600.0 >= dt && p[7] * 3600.0 >= dt && p[8] * 3600.0 >= dt && p[9] * 3600.0 >= dt);
    const bool guard = !(p[4] >= 0.0 && p[4] <= 0.5 && p[1] >= 0.0 && p[5] > 0.0);
    // dt / RK > 2: the river's explicit update multiplies a perturbation by |1 - dt/RK| > 1 on every step its 95 % rule
    // does not fire, and the rule itself only damps it to 5 % (structure.py:490-498) -- between dt/RK = 10 and 20 the
    // two alternate without settling and the last bit of the inflow decides the discharge.  The five catchment
    // reservoirs are no concern at any dt/k: their clamp at zero (:429-450) forgets the perturbation altogether
    // (measured to dt/k = 200: <= 3e-10 of the reference; tools/debug/illcond_err.py).
#ifdef SMART_NO_ILLCOND // measurement builds only (tools/debug/illcond_err.py): what the fast arithmetic does on such rows
    const bool unstable = false;
#else
    const bool unstable = !(p[9] * 3600.0 >= 0.5 * dt);
#endif
    // a parameter that is NaN or infinite: the reference's own operation order, compiled with NaNs honoured (the fast
    // kernels are not: -fno-honor-nans), decides what comes out -- tested on the bit patterns for that reason
    bool wild = false;
#pragma unroll
    for (int i = 0; i < 10; ++i)
        wild = wild || (__builtin_bit_cast(unsigned long long, p[i]) & 0x7ff0000000000000ull) == 0x7ff0000000000000ull;
    // ... or a share that is none: a drain fraction D or an overland share H outside [0, 1], a negative rain factor T hand
    // the reservoirs NEGATIVE inflows, and what the reference's clamps make of those (structure.py:429-450) only its
    // own operation order reproduces (round 4: D = 300 with a soil of half a millimetre, fast mode off by a factor)
    wild = wild || !(p[3] >= 0.0 && p[3] <= 1.0) || !(p[2] >= 0.0 && p[2] <= 1.0) || !(p[0] >= 0.0);
    // ... or a residence time that is none: with k <= 0 a reservoir grows by itself, step after step -- the reference's
    // numbers explode in the reference's own way, and the stiff variant's clamps know nothing of it
    wild = wild || !(p[6] > 0.0 && p[7] > 0.0 && p[8] > 0.0 && p[9] > 0.0);
    // ... or a soil without capacity (Z <= 0: levels are quotients by it -- 0 / 0, x / 0 and what the reference's compares
    // make of the NaNs and infinities)
    wild = wild || !(p[5] > 0.0);
    // ... or a discharge orders below what the rain would make (T < 0.2, or a soil of more than a metre that takes
    // all of it): the fast arithmetic's sums carry the rain's last place -- 1e-15 mm a step -- which is 1e-6 of such
    // a row's small values
    wild = wild || !(p[0] >= 0.2) || !(p[5] <= 1.0e3);
    // ... or a soil of less than a millimetre, which every rainy step overflows (met with C < 0: a NaN on one side only)
    wild = wild || !(p[5] >= 1.0);
    // ... and a caller's INITIAL states the fast arithmetic is not made for: a NaN, an infinity, a negative volume (the
    // reference's clamps and compares decide what follows), or soil so far above its capacity that s' = S tot / Z
    // starts beyond the 0.5 the guard class stops at (tot / Z <= 1 from the first wet step on: the filling clamps)
    if (a.initial) {
        const double *ip = a.initial + (catchment * a.N + n) * 12;
        double lay = 0.0;
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const unsigned long long u = __builtin_bit_cast(unsigned long long, ip[i]);
            wild = wild || (u & 0x7ff0000000000000ull) == 0x7ff0000000000000ull || u > 0x8000000000000000ull;
            if (i >= 5 && i < 11)
                lay += ip[i];
        }
        if (!wild) { // (finite numbers in; the quotient can still be a NaN -- area 0, 0 * inf -- and this function is
                     // compiled with and without -fno-honor-nans: the NaN is tested on its bit pattern, like the host does
                     // by `~(s <= 0.5)` on honest IEEE arithmetic)
            const double fill = (lay / a.area[catchment] * 1e3) / p[5]; // tot / Z of the first step
            const double s_init = p[4] * fill;
            // ... or that the overland share H tot / Z of a rainy step's excess starts beyond one: the reference then hands
            // a NEGATIVE excess to the filling and takes it out of the top layer (structure.py:363-370), which may go
            // below zero -- where its guard `leak < level` lets nothing leak and the fast arithmetic's unguarded leaks
            // would (round 4; the same hole the fuzzer found in the reciprocal path, smart_literal_model.h)
            const double h_init = p[2] * fill;
#ifdef SMART_NO_HINIT // measurement builds only: what the fast arithmetic makes of such a start
            wild = is_nan_bits(s_init) || !(s_init <= 0.5);
            (void)h_init;
#else
            wild = is_nan_bits(s_init) || !(s_init <= 0.5) || is_nan_bits(h_init) || !(h_init <= 1.0);
#endif
        }
    }
    const bool any_stiff = __builtin_amdgcn_ballot_w64(stiff) != 0;
    const bool any_guard = __builtin_amdgcn_ballot_w64(guard) != 0;
    const bool any_unstable = __builtin_amdgcn_ballot_w64(unstable || wild) != 0;
    return any_unstable ? 3 : (any_guard ? 2 : (any_stiff ? 1 : 0));
}

} // namespace smart
