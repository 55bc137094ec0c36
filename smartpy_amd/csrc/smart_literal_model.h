// smart_literal_model.h -- the reference's arithmetic, operation for operation (structure.py:267-503).
//
// Every function body starts with `#pragma clang fp contract(off)`: each +, -, *, / rounds once, in the order
// CPython evaluates the reference, with IEEE-754 correctly rounded fp64 division, whatever -ffp-contract the
// including translation unit is compiled with.  The one deliberate difference is s' ** i (structure.py:382,396):
// CPython calls libm pow(), whose last bit depends on the host's glibc build; here it is the product chain
// ((s*s)*s)*... .  With the CPU oracle configured the same way this model is bit-identical to it.
//
// Used by the literal kernel (smart_literal.hip) and, inside the fast kernel, for wavefronts that hold a sample
// with dt / RK > 2: there the river's explicit Euler update amplifies any rounding difference by |1 - dt/RK| on
// every step its 95 % rule does not fire, so only the reference's own operation order reproduces its discharge
// ("Ill-conditioned samples").
//
// RECIP (the ill-conditioned rows of the fast mode, smart_fast_illcond): 31 of a step's 39 divisions are by one of ten
// per-sample constants (area, Z, dt, the four routing constants) or by 1e3 / 2 .. 6, and an IEEE division is ten
// instructions on gfx950 (v_div_scale x2, v_rcp, four refinement FMAs, v_div_fmas, v_div_fixup) -- half of the literal
// step.  With y = RN(1 / b) kept per sample (one true division at set-up), Markstein's correction step
//     q0 = RN(a y);   r = RN(a - b q0)  [exact: one FMA];   q = RN(q0 + r y)
// returns the correctly rounded quotient RN(a / b) -- THE SAME BITS as the division -- for every a, unless b's
// significand is all ones (P. Markstein, IBM J. Res. Dev. 34 (1990), Theorem 4.2; the tail of hipcc's own division
// sequence is this step) or the computation leaves the normal range (r is exact only while a - b q0 is representable:
// |a| >= 2^-969 is enough).  The theorem wants y correctly rounded (it is: one true division) and q0 within an ulp of
// a / b; RN(a y) is NOT always that close (two roundings of relative size 2^-53 are up to two ulps of a quotient whose
// significand is next to 2: the remainder then needs 54 bits and the FMA rounds it) -- the claim of bit identity
// rests on the theorem where its hypothesis holds AND on comparisons: tests/test_host_logic.py replays the three
// operations in exact rational arithmetic on 1.5e5 pairs drawn to stress them (significands next to 1 and 2, divisors
// of the model's kind): the remainder is inexact on 0.9 % of them and the quotient is the division's on all; and
// tools/debug/recip_bits.py compared 5.5e7 kernel outputs with the true divisions (profiles/r03_recip_bits.txt): 0
// differ.  Both conditions are checked, wave-uniformly: the divisors once per run ([2^-500, 2^500]), the twelve
// states and the step's forcing once per chunk of four steps against [2^-400, 2^400) (a state shrinks by 2^-53 per
// step at most: (1 - dt/k) of a double dt/k < 1) -- every quotient and every remainder then stays a normal number;
// a wave that fails takes the true divisions for that chunk.  What is
// NOT checked are numerators formed inside the step from products with tiny factors (a leak l s'^6 with s' ~ 1e-50):
// a quotient of a numerator below 2^-969 may differ in its last bit -- of a number that is zero to any hydrologist.
// tests/test_gpu_parity.py keeps comparing the kernel with the literal kernel (true divisions throughout) bit for bit.
#pragma once

#include "smart_device.h"

namespace smart {

template <bool RECIP>
struct LiteralModelT {
    static constexpr bool kExactDivide = true;
    static constexpr bool kBalanceSums = false;
    static constexpr bool kTracksOutputs = true; // out[] holds the seven outputs of the last step taken
    static constexpr bool kChunkHook = RECIP;    // time_loop_chunked() calls begin_chunk() ahead of every chunk

    double area, dt;
    double pT, pC, pH, pD, pS, pZ, sk, fk, gk, rk;
    double v_ove, v_dra, v_int, v_sgw, v_dgw, v_ly[6], v_riv;
    double out[7];
    double q_out, q_in, q_gw;

    __device__ void setup(double area_m2, double delta, const double *p)
    {
#pragma clang fp contract(off)
        area = area_m2;
        dt = delta;
        pT = p[0];
        pC = p[1];
        pH = p[2];
        pD = p[3];
        pS = p[4];
        pZ = p[5];
        sk = p[6] * 3600.0; // structure.py:320-322
        fk = p[7] * 3600.0;
        gk = p[8] * 3600.0;
        rk = p[9] * 3600.0; // structure.py:482
#pragma unroll
        for (int i = 0; i < 7; ++i)
            out[i] = 0.0;
        if constexpr (RECIP) {
            y_area = 1.0 / area, y_z = 1.0 / pZ, y_dt = 1.0 / dt;
            y_sk = 1.0 / sk, y_fk = 1.0 / fk, y_gk = 1.0 / gk, y_rk = 1.0 / rk;
            // a divisor is fit for the correction step if it is a positive normal number within [2^-500, 2^500] whose
            // significand is not all ones
            auto fit = [](double b) {
                const unsigned long long u = __builtin_bit_cast(unsigned long long, b);
                const unsigned long long frac = u & 0x000fffffffffffffull;
                return u - 0x20b0000000000000ull < 0x3e80000000000000ull && frac != 0x000fffffffffffffull;
            };
            // ... and the other five parameters finite: the identities of the reciprocal path (clamps and hand-downs as
            // maxima / minima, "d = 0.0" as C (d - d)) hold for numbers, not for a NaN that has to come out as one
            auto finite = [](double x) {
                return (__builtin_bit_cast(unsigned long long, x) & 0x7ff0000000000000ull) != 0x7ff0000000000000ull;
            };
            const bool ok = fit(area) && fit(pZ) && fit(dt) && fit(sk) && fit(fk) && fit(gk) && fit(rk) && finite(pT) &&
                            finite(pC) && finite(pH) && finite(pD) && finite(pS);
            divisors_fit = __builtin_amdgcn_ballot_w64(!ok) == 0;
            quick = false;
        }
    }

    // ---- RECIP: division by a per-sample constant b with y = RN(1 / b) at hand -------------------------------------
    double y_area, y_z, y_dt, y_sk, y_fk, y_gk, y_rk;
    bool divisors_fit, quick; // wave-uniform: every divisor of the wave is fit; this chunk may use the correction step

    template <bool Q>
    __device__ __forceinline__ static double dv(double a, double b, double y)
    {
#pragma clang fp contract(off)
        if constexpr (Q) {
            const double q0 = a * y;
            const double r = __builtin_fma(-b, q0, a);
            return __builtin_fma(r, y, q0);
        } else {
            return a / b;
        }
    }

    // 0 or within [2^-400, 2^400): with the divisors in [2^-500, 2^500] every quotient a / b of the step, and the
    // remainder a - b q0 of its correction step (~ 2^-53 a), stays a NORMAL number (>= 2^-953) -- the exact-remainder
    // assumption of the correction step, and what keeps subnormal layers (whose unguarded leak would round differently)
    // off this path.  (Round 3 admitted [2^-800, 2^800): quotients down to 2^-1300, i.e. subnormal or zero -- advisor.)
    __device__ __forceinline__ static bool in_range(double x)
    {
        const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
#ifdef SMART_RECIP_RANGE_R03 // (debug builds only: round 3's [2^-800, 2^800))
        return u == 0 || u - 0x0df0000000000000ull < 0x6400000000000000ull;
#endif
        return u == 0 || u - 0x26f0000000000000ull < 0x3200000000000000ull;
    }

    // ahead of a chunk of (at most) four steps, with their forcing
    __device__ __forceinline__ void begin_chunk(const double2 *f4, int n)
    {
        if constexpr (RECIP) {
            bool ok = in_range(v_ove) && in_range(v_dra) && in_range(v_int) && in_range(v_sgw) && in_range(v_dgw) &&
                      in_range(v_riv);
#pragma unroll
            for (int i = 0; i < 6; ++i)
                ok = ok && in_range(v_ly[i]);
            bool forcing_ok = true; // (scalar unit: the forcing is wave-uniform)
            for (int j = 0; j < n; ++j)
                forcing_ok = forcing_ok && in_range(f4[j].x) && in_range(f4[j].y);
            quick = divisors_fit && forcing_ok && __builtin_amdgcn_ballot_w64(!ok) == 0;
        }
    }

    __device__ void set_states(const double *st)
    {
        v_ove = st[0];
        v_dra = st[1];
        v_int = st[2];
        v_sgw = st[3];
        v_dgw = st[4];
#pragma unroll
        for (int i = 0; i < 6; ++i)
            v_ly[i] = st[5 + i];
        v_riv = st[11];
    }

    __device__ void flows_of_next_step(double, double, double, double *) const {}

    __device__ void get_vars(double *v, const double * = nullptr) const
    {
#pragma unroll
        for (int i = 0; i < 7; ++i)
            v[i] = out[i];
        v[7] = v_ove;
        v[8] = v_dra;
        v[9] = v_int;
        v[10] = v_sgw;
        v[11] = v_dgw;
#pragma unroll
        for (int i = 0; i < 6; ++i)
            v[12 + i] = v_ly[i];
        v[18] = v_riv;
    }

    // one linear reservoir, structure.py:427-450
    // run_one_step_river, structure.py:461-503, with rk already in seconds: outflow of the step, v updated
    __device__ static double river(double dt, double q_in, double rk, double &v)
    {
        return river_q<false>(dt, q_in, rk, v, 0.0, 0.0);
    }

    template <bool Q>
    __device__ static double river_q(double dt, double q_in, double rk, double &v, double y_rk, double y_dt)
    {
#pragma clang fp contract(off)
        double q = dv<Q>(v, rk, y_rk); // :487
        const double v_old = v;
        const double tmp = v_old + (q_in - q) * dt; // :490
        if (tmp < 0.0) {                            // :492-496
            q = 0.95 * (q_in + dv<Q>(v_old, dt, y_dt));
            v += (q_in - q) * dt;
        } else {
            v = tmp; // :498
        }
        return q;
    }

    template <bool Q>
    __device__ static double route(double &v, double k, double y_k, double x_mm, double area, double dt)
    {
#pragma clang fp contract(off)
        const double q = dv<Q>(v, k, y_k);
        v += (dv<Q>(x_mm, 1e3, 1e-3) * area) - (q * dt);
        if constexpr (Q) {
            // "if V < 0: V = 0" (:429-450) as a maximum: one instruction instead of a compare and two selects, the same
            // bits -- V is never -0.0 (it starts at +0 or above, and x + (-x) = +0) and never NaN on this path
            v = __builtin_fmax(v, 0.0);
        } else {
            if (v < 0.0)
                v = 0.0;
        }
        return q;
    }

    // time_loop() asks for the excess ahead of the step; the literal step recomputes it in the reference's order
    __device__ double excess(double rain_in, double peva_in) const
    {
#pragma clang fp contract(off)
        return rain_in * pT - peva_in;
    }

    __device__ void step(double rain_in, double peva_in, double /*ex*/, double &acc, double &num, double &den)
    {
        if constexpr (RECIP) {
            if (__builtin_expect(quick, 1)) { // wave-uniform; the other form out of the way of the loop's instruction stream
                step_q<true>(rain_in, peva_in, acc, num, den);
                return;
            }
        }
        step_q<false>(rain_in, peva_in, acc, num, den);
    }

    template <bool Q>
    __device__ __forceinline__ void step_q(double rain_in, double peva_in, double &acc, double &num, double &den)
    {
#pragma clang fp contract(off)
        const double z = pZ / 6.0; // structure.py:329-337
        double l[6];
#pragma unroll
        for (int i = 0; i < 6; ++i)
            l[i] = dv<Q>(v_ly[i], area, y_area) * 1e3; // :339-347
        double tot = 0.0 + 0.0;          // Python's sum() over [0.0, l1..l6] (:350)
#pragma unroll
        for (int i = 0; i < 6; ++i)
            tot = tot + l[i];

        const double rain = rain_in * pT; // :353
        double ex = rain - peva_in;       // :355
        double aeva = 0.0;
        double of, df, inf, sh, dp;

        if (ex >= 0.0) { // :359
            aeva += peva_in;
            const double hp = pH * dv<Q>(tot, pZ, y_z); // :363
            of = hp * ex;
            ex -= of;
            [[maybe_unused]] const double ex_in = ex; // what the filling starts from: NEGATIVE when H tot / Z > 1 (soil above capacity)
#pragma unroll
            for (int i = 0; i < 6; ++i) { // :367-374
                const double sp = z - l[i];
                if constexpr (Q) {
                    // what the layer takes is min(ex, sp) either way, and "ex = 0.0" is ex - ex: the excess handed down
                    // without its two selects (the level keeps its own: l + (z - l) need not round to z)
                    const double put = __builtin_fmin(ex, sp);
                    l[i] = ex <= sp ? l[i] + ex : z;
                    ex = ex - put;
                } else if (ex <= sp) {
                    l[i] += ex;
                    ex = 0.0;
                } else {
                    l[i] = z;
                    ex -= sp;
                }
            }
            df = pD * ex;                      // :376
            inf = (1.0 - pD) * ex;             // :377
            const double s1 = pS * dv<Q>(tot, pZ, y_z); // :379
            double pw[6];                      // s1 ** (i + 1) as a product chain
            pw[0] = s1;
#pragma unroll
            for (int i = 1; i < 6; ++i)
                pw[i] = pw[i - 1] * s1;
            // The guard `if lk < l` of the three leak passes (:383, :390, :397) is false only for an empty layer as long
            // as the factor is well below one (lk = RN(l f) < l for every normal l > 0 when f <= 0.75), and for an
            // empty layer the unguarded update adds and takes a zero: the same bits without a compare and two selects
            // per layer and pass.  s' <= 0.75 bounds every factor (s'^i, s' / i); decided per step for the lanes on
            // the wet side, wave-uniformly; the reciprocal path only (its range check keeps subnormal layers, where
            // RN(l f) can equal l, on the other one).  Measured: 4.60 -> 4.38 ms for config 2.
            // (Also measured, and slower, 4.68 ms: running only the wet or only the dry side when every lane of the
            // wave agrees, instead of the per-lane if / else.)
            // ... and as long as no level is NEGATIVE: for l < 0 the reference's guard is false (l f > l) and nothing leaks,
            // where the unguarded update would take l f all the same.  Levels are >= 0 when a chunk starts (in_range) and
            // stay so through the evaporation cascade and the leaks; the one way below zero is a negative excess handed to
            // the filling -- the overland share H tot / Z of the excess exceeds one when the soil stands above its
            // capacity (a negative C, which class 3 takes too when dt / RK > 2, or a caller's initial state) -- and the
            // reference then takes it out of the top layer (:367-370, `lvl + excess`).  Found by the fuzzer in round 4
            // (seeds 9001, 9040, 9055 of tools/debug/fuzz_wide.py: one row in 190 off by 1e-5, amplified by the river):
            // a step that meets one keeps its guards, and so does the rest of its chunk (begin_chunk() looks again).
            bool unguarded = false;
            if constexpr (Q) {
                unguarded = __builtin_amdgcn_ballot_w64(!(s1 <= 0.75 && s1 >= 0.0 && ex_in >= 0.0)) == 0;
                if (!unguarded)
                    quick = false;
            }
            auto take = [](double &level, double f, double &into) {
                const double lk = level * f;
                if (lk < level) {
                    into += lk;
                    level -= lk;
                }
            };
            if (__builtin_expect(unguarded, 1)) { // (two copies of the passes, one branch: the flag is wave-uniform)
#pragma unroll
                for (int i = 0; i < 6; ++i) { // :381-385
                    const double lk = l[i] * pw[i];
                    inf += lk;
                    l[i] -= lk;
                }
                sh = 0.0;
#pragma unroll
                for (int i = 0; i < 6; ++i) { // :387-392
                    constexpr double by[6] = {1.0, 1.0 / 2.0, 1.0 / 3.0, 1.0 / 4.0, 1.0 / 5.0, 1.0 / 6.0};
                    const double lk = l[i] * (i == 0 ? s1 : dv<Q>(s1, (double)(i + 1), by[i]));
                    sh += lk;
                    l[i] -= lk;
                }
                dp = 0.0;
#pragma unroll
                for (int i = 5; i >= 0; --i) { // :394-399
                    const double lk = l[i] * pw[5 - i];
                    dp += lk;
                    l[i] -= lk;
                }
            } else {
#pragma unroll
                for (int i = 0; i < 6; ++i) // :381-385
                    take(l[i], pw[i], inf);
                sh = 0.0;
#pragma unroll
                for (int i = 0; i < 6; ++i) { // :387-392
                    constexpr double by[6] = {1.0, 1.0 / 2.0, 1.0 / 3.0, 1.0 / 4.0, 1.0 / 5.0, 1.0 / 6.0}; // RN(1 / (i + 1))
                    take(l[i], i == 0 ? s1 : dv<Q>(s1, (double)(i + 1), by[i]), sh); // s1 / 1 is s1
                }
                dp = 0.0;
#pragma unroll
                for (int i = 5; i >= 0; --i) // :394-399, bottom layer first, exponent 7 - layer
                    take(l[i], pw[5 - i], dp);
            }
        } else { // :400
            of = 0.0;
            df = 0.0;
            inf = 0.0;
            sh = 0.0;
            dp = 0.0;
            double d = ex * (-1.0); // :407
            aeva += rain;
#pragma unroll
            for (int i = 0; i < 6; ++i) { // :409-419
                if constexpr (Q) {
                    // both sides of the reference's branch take min(l, d) from the layer and add it to the evaporation;
                    // "l = 0.0" is l - l, "d = 0.0" is C (d - d) for a finite C (checked once per run, with the
                    // divisors): the same bits without a compare and three selects per layer
                    const double take = __builtin_fmin(l[i], d);
                    l[i] -= take;
                    aeva += take;
                    d = pC * (d - take);
                } else if (l[i] >= d) {
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

        out[0] = dv<Q>(dv<Q>(aeva, 1e3, 1e-3) * area, dt, y_dt); // :424
        out[1] = route<Q>(v_ove, sk, y_sk, of, area, dt);
        out[2] = route<Q>(v_dra, sk, y_sk, df, area, dt);
        out[3] = route<Q>(v_int, fk, y_fk, inf, area, dt);
        out[4] = route<Q>(v_sgw, gk, y_gk, sh, area, dt);
        out[5] = route<Q>(v_dgw, gk, y_gk, dp, area, dt);
#pragma unroll
        for (int i = 0; i < 6; ++i)
            v_ly[i] = dv<Q>(l[i], 1e3, 1e-3) * area; // :456-457

        // river, structure.py:487-498; inflow summed left to right (:254)
        q_in = out[1] + out[2] + out[3] + out[4] + out[5];
        q_gw = out[4] + out[5];
        const double q = river_q<Q>(dt, q_in, rk, v_riv, y_rk, y_dt);
        out[6] = q;
        q_out = q;
        acc += q;
        num += q_gw;
        den += q_in;
    }
};

using LiteralModel = LiteralModelT<false>;      // the literal kernel: true divisions, nothing to check
using LiteralModelRecip = LiteralModelT<true>;  // the ill-conditioned rows of the fast mode

} // namespace smart
