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
#pragma once

#include "smart_device.h"

namespace smart {

struct LiteralModel {
    static constexpr bool kExactDivide = true;
    static constexpr bool kBalanceSums = false;
    static constexpr bool kTracksOutputs = true; // out[] holds the seven outputs of the last step taken

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
#pragma clang fp contract(off)
        double q = v / rk; // :487
        const double v_old = v;
        const double tmp = v_old + (q_in - q) * dt; // :490
        if (tmp < 0.0) {                            // :492-496
            q = 0.95 * (q_in + v_old / dt);
            v += (q_in - q) * dt;
        } else {
            v = tmp; // :498
        }
        return q;
    }

    __device__ static double route(double &v, double k, double x_mm, double area, double dt)
    {
#pragma clang fp contract(off)
        const double q = v / k;
        v += (x_mm / 1e3 * area) - (q * dt);
        if (v < 0.0)
            v = 0.0;
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
#pragma clang fp contract(off)
        const double z = pZ / 6.0; // structure.py:329-337
        double l[6];
#pragma unroll
        for (int i = 0; i < 6; ++i)
            l[i] = v_ly[i] / area * 1e3; // :339-347
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
            const double hp = pH * (tot / pZ); // :363
            of = hp * ex;
            ex -= of;
#pragma unroll
            for (int i = 0; i < 6; ++i) { // :367-374
                const double sp = z - l[i];
                if (ex <= sp) {
                    l[i] += ex;
                    ex = 0.0;
                } else {
                    l[i] = z;
                    ex -= sp;
                }
            }
            df = pD * ex;                      // :376
            inf = (1.0 - pD) * ex;             // :377
            const double s1 = pS * (tot / pZ); // :379
            double pw[6];                      // s1 ** (i + 1) as a product chain
            pw[0] = s1;
#pragma unroll
            for (int i = 1; i < 6; ++i)
                pw[i] = pw[i - 1] * s1;
#pragma unroll
            for (int i = 0; i < 6; ++i) { // :381-385
                const double lk = l[i] * pw[i];
                if (lk < l[i]) {
                    inf += lk;
                    l[i] -= lk;
                }
            }
            sh = 0.0;
#pragma unroll
            for (int i = 0; i < 6; ++i) { // :387-392
                const double lk = l[i] * (s1 / (double)(i + 1));
                if (lk < l[i]) {
                    sh += lk;
                    l[i] -= lk;
                }
            }
            dp = 0.0;
#pragma unroll
            for (int i = 5; i >= 0; --i) { // :394-399, bottom layer first, exponent 7 - layer
                const double lk = l[i] * pw[5 - i];
                if (lk < l[i]) {
                    dp += lk;
                    l[i] -= lk;
                }
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

        out[0] = aeva / 1e3 * area / dt; // :424
        out[1] = route(v_ove, sk, of, area, dt);
        out[2] = route(v_dra, sk, df, area, dt);
        out[3] = route(v_int, fk, inf, area, dt);
        out[4] = route(v_sgw, gk, sh, area, dt);
        out[5] = route(v_dgw, gk, dp, area, dt);
#pragma unroll
        for (int i = 0; i < 6; ++i)
            v_ly[i] = l[i] / 1e3 * area; // :456-457

        // river, structure.py:487-498; inflow summed left to right (:254)
        q_in = out[1] + out[2] + out[3] + out[4] + out[5];
        q_gw = out[4] + out[5];
        const double q = river(dt, q_in, rk, v_riv);
        out[6] = q;
        q_out = q;
        acc += q;
        num += q_gw;
        den += q_in;
    }
};

} // namespace smart
