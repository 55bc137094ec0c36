// smart_literal_lanes.h -- the literal step (smart_literal_model.h, structure.py:267-503) with ONE SAMPLE SPREAD OVER THE
// SIXTEEN LANES OF A DPP ROW instead of one sample per lane: four samples per wavefront.
//
// Why (round 5).  Wherever the reference's own operation order is wanted -- the ill-conditioned rows of a daily ensemble
// (dt / RK > 2: smart_fast_illcond) and the smartcpp.allsteps stand-in, one sample per call -- there are far fewer
// samples than the chip has SIMDs, a wavefront is alone on its SIMD, and a lone wavefront issues one instruction of any
// kind per ~5 cycles whatever its lanes hold (profiles/r04_microbench_lone.txt): the time of a step is its INSTRUCTION
// COUNT, ~490 for the lane-per-sample form.  Most of those are the same operation repeated for the six soil layers and
// the five reservoirs.  Here the six layers sit in six lanes (one instruction for all of them), the five reservoirs in
// five lanes, and the sums the reference takes in a fixed order -- sum() over the layers (:350), the three leak totals
// (:381-399), the evaporation (:409-419), the river's inflow (:254) -- are chains of
//     v_fmac_f64_dpp acc, x, one  row_newbcast:k        acc = fma(x[lane k of the row], 1.0, acc)
// one instruction per term, in the reference's order, the sum arriving in every lane of the row.  (The product with
// 1.0 is exact, so the fma rounds once, like the addition: the same bits.  gfx950's 64-bit DPP knows row_newbcast
// only, and only for v_mov_b64 and v_fmac_f64 among the instructions of use here -- which is exactly enough.)
// The two cascades that hand a remainder from layer to layer (filling :367-374, evaporation :409-419) run as chains
// over the row too, with per-lane constants that switch a lane's copy of the chain off at its own layer:
//     e_{k+1} = max(fma(sp[k], m, e_k), 0)      m = -1.0 in the lanes below layer k, 0.0 in the others
// so lane i ends up holding what arrives at ITS layer, and the lanes that are no layer the final remainder.
// Same operations on the same operands as LiteralModelT<true>'s reciprocal path, hence the same bits -- which
// tests/test_gpu_parity.py holds against the literal kernel and the oracle as before.  What the reciprocal path
// checks, this one checks (divisors once per run, states and forcing once per chunk, s' and the excess per step); a
// step that fails a check gathers the row's states into LiteralModelT's lane-per-sample form, takes ITS step with true
// divisions and guards, and scatters them back: nothing is approximated anywhere.
//
// Lane roles within a row (r = lane & 15):
//   soil layers        r = 0 .. 5   (LY: the layer's volume; l, sp, leaks, takes: one register for all six)
//   reservoirs (V)     sgw r = 0, dgw r = 4, int r = 8, ove r = 12, dra r = 13 -- one per DPP bank for the three that are
//                      fed by a chain (bank_mask lets a chain's additions land in one bank only: sh -> bank 0, dp ->
//                      bank 1, inf -> bank 2), ove and dra, fed by one product each, share bank 3
//   river, parameters  every lane of the row holds the same value
// Lanes without a role compute on benign values (V of a lane that is no reservoir only collects what its bank's chain
// adds, behind a residence time of 2^500 s); nothing is ever broadcast from them and the range checks skip them.
#pragma once

#include "smart_literal_model.h"

namespace smart {

#define SMART_L_DPP(k) " row_newbcast:" #k " row_mask:0xf bank_mask:0xf\n\t"
#define SMART_L_DPPB(k, b) " row_newbcast:" #k " row_mask:0xf bank_mask:" #b "\n\t"

// The guarded form of one step for the row form's fallback: LiteralModel's own step (true divisions, every guard), on
// a block of doubles in memory -- in: [0..9] the ten parameters as sampled, [10] area, [11] dt, [12..23] the twelve
// states, [24] rain, [25] peva; out: [12..23] the states, [26..32] the seven outputs, [33] q_in, [34] q_gw.  A real
// call (noinline), so that its registers are its own: inlined, the ~180 VGPRs of the literal step on top of the row
// form's constants spilled into the hot loop (round 5: 117 ms per smartcpp.allsteps call instead of 61).  Every lane
// of a row computes the same numbers.
__device__ __attribute__((noinline, cold)) static void literal_guarded_step(double *io)
{
    LiteralModel m;
    m.setup(io[10], io[11], io);
    m.set_states(io + 12);
    double acc = 0.0, num = 0.0, den = 0.0;
    m.step(io[24], io[25], 0.0, acc, num, den);
    double v[19];
    m.get_vars(v);
#pragma unroll
    for (int i = 0; i < 12; ++i)
        io[12 + i] = v[7 + i];
#pragma unroll
    for (int i = 0; i < 7; ++i)
        io[26 + i] = v[i];
    io[33] = m.q_in;
    io[34] = m.q_gw;
}

struct LiteralLanesModel {
    static constexpr int kLanesPerSample = 16;
    static constexpr bool kExactDivide = true;
    static constexpr bool kBalanceSums = false;
    static constexpr bool kTracksOutputs = true;
    static constexpr bool kChunkHook = true;
    static constexpr bool kChunkModes = true; // time_loop_chunked() picks the form of a chunk's steps on `quick`
    using Slow = LiteralModelT<true>;

    // lanes of the row that hold the reservoirs (out[1..5] = ove, dra, int, sgw, dgw)
    static constexpr int kOve = 12, kDra = 13, kInt = 8, kSgw = 0, kDgw = 4;

    struct {           // parameters and reciprocals, every lane of the row the same
        double area, dt, pT, pC, pH, pD, pS, pZ, rk;
        double y_area, y_z, y_dt, y_rk;
        bool divisors_fit;
    } s;
    double raw_k[4];   // SK, FK, GK, RK as sampled, hours (the guarded form's setup starts from them)
    double z, omd;     // pZ / 6, 1 - pD
    double zcap;       // z (1 + 2^-40): a layer 'within capacity' as far as begin_chunk is concerned
    double one;        // 1.0 in a register (src1 of v_fmac_f64_dpp is a VGPR)
    // per-lane constants
    double kres;       // residence time of this lane's reservoir (2^500 s in the lanes that hold none)
    double m[6];       // filling / evaporation chains: -1.0 while the chain is above this lane's layer, then 0.0
    double cm[6];      // evaporation chain: pC while above this lane's layer, then 1.0
    double w_of;       // 0.0 in the overland reservoir's lane (its copy of the excess keeps the overland share), else 1.0
    double sx, kx;     // the factor that makes this lane's inflow of its copy of the excess: hp * sx + kx
    double dk[5], ek[5]; // power chain: step k multiplies by s' * dk + ek (s' or 1.0)
    double nby, by;    // layer r: r + 1 and RN(1 / (r + 1)) (the shallow-groundwater pass divides s' by it)
    bool is_layer, is_res;
    int role;
    // states
    double LY, V, VRIV;
    // outputs of the last step
    double out0, outq, out6;
    double aeva_last;  // actual evaporation [mm] of the last quick step: out[0] = (aeva / 1e3) area / dt, on demand
    double q_out, q_in, q_gw;
    bool quick;
#ifdef SMART_LANES_COUNT
    int n_slow = 0; // (microbenchmark builds: steps that took the guarded form)
    int n_why[6] = {0, 0, 0, 0, 0, 0}; // chunks refused for: divisors, forcing, a layer's range, a layer above capacity, the river
#endif

    __device__ void setup(double area_m2, double delta, const double *p)
    {
#pragma clang fp contract(off)
        s.area = area_m2, s.dt = delta;
        s.pT = p[0], s.pC = p[1], s.pH = p[2], s.pD = p[3], s.pS = p[4], s.pZ = p[5];
        const double sk = p[6] * 3600.0, fk = p[7] * 3600.0, gk = p[8] * 3600.0; // structure.py:320-322
        s.rk = p[9] * 3600.0;                                                     // structure.py:482
#pragma unroll
        for (int i = 0; i < 4; ++i)
            raw_k[i] = p[6 + i];
        s.y_area = 1.0 / s.area, s.y_z = 1.0 / s.pZ, s.y_dt = 1.0 / s.dt, s.y_rk = 1.0 / s.rk;
        role = (int)(threadIdx.x & 15u);
        const int r = role;
        is_layer = r < 6;
        is_res = r == kOve || r == kDra || r == kInt || r == kSgw || r == kDgw;
        z = s.pZ / 6.0;
        omd = 1.0 - s.pD;
        zcap = z * (1.0 + 0x1p-40);
        asm volatile("v_mov_b64 %0, 1.0" : "=v"(one));
        const double big = 0x1p+500;
        kres = (r == kOve || r == kDra) ? sk : (r == kInt ? fk : ((r == kSgw || r == kDgw) ? gk : big));
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const bool above = r >= 6 ? r != kOve : k < r;
            m[k] = above ? -1.0 : 0.0;
            cm[k] = above ? s.pC : 1.0;
        }
        w_of = r == kOve ? 0.0 : 1.0;
        sx = r == kOve ? 1.0 : 0.0;
        kx = r == kDra ? s.pD : (r == kInt ? omd : 0.0);
#pragma unroll
        for (int k = 1; k <= 5; ++k) {
            // lanes 0..5: s'^(r + 1) for the interflow pass; lanes 8..13: s'^(6 - j) for layer j = r - 8 (the deep pass),
            // shifted down by eight lanes afterwards
            const bool mul = r < 6 ? k <= r : (r >= 8 && r < 14 ? k <= 5 - (r - 8) : false);
            dk[k - 1] = mul ? 1.0 : 0.0;
            ek[k - 1] = mul ? 0.0 : 1.0;
        }
        nby = is_layer ? (double)(r + 1) : 1.0;
        by = 1.0 / nby;
        out0 = outq = out6 = aeva_last = 0.0;
        q_out = q_in = q_gw = 0.0;
        quick = false;
        // The conditions of LiteralModelT<true>'s reciprocal path: a divisor is fit for the correction step if it is a
        // positive normal number within [2^-500, 2^500] whose significand is not all ones; the other parameters finite
        auto fit = [](double b) {
            const unsigned long long u = __builtin_bit_cast(unsigned long long, b);
            const unsigned long long frac = u & 0x000fffffffffffffull;
            return u - 0x20b0000000000000ull < 0x3e80000000000000ull && frac != 0x000fffffffffffffull;
        };
        auto finite = [](double x) {
            return (__builtin_bit_cast(unsigned long long, x) & 0x7ff0000000000000ull) != 0x7ff0000000000000ull;
        };
        bool ok = fit(s.area) && fit(s.pZ) && fit(s.dt) && fit(sk) && fit(fk) && fit(gk) && fit(s.rk) && finite(s.pT) &&
                  finite(s.pC) && finite(s.pH) && finite(s.pD) && finite(s.pS);
        // ... and what the row form adds: a decay C >= 0 (the max-form of the evaporation chain: with C < 0 the handed-down
        // demand changes sign and the zeros' signs with it); a rain factor T >= 0 and a soil outflow coefficient S >= 0
        // (rain and s' are then +0 or positive, which spares the step two additions of 0.0 and a compare); an overland
        // share H <= 0.99 and S <= 0.74 (with layers within capacity the leak factors then stay below the 0.75 up to which
        // the unguarded leak passes are the guarded ones, and the excess stays >= 0: see step)
        ok = ok && s.pC >= 0.0 && s.pT >= 0.0 && s.pS >= 0.0 && s.pS <= 0.74 && s.pH <= 0.99;
        s.divisors_fit = __builtin_amdgcn_ballot_w64(!ok) == 0;
    }

    __device__ void set_states(const double *st)
    {
        const int r = role;
        V = r == kOve ? st[0] : (r == kDra ? st[1] : (r == kInt ? st[2] : (r == kSgw ? st[3] : (r == kDgw ? st[4] : 0.0))));
        LY = r == 0 ? st[5] : (r == 1 ? st[6] : (r == 2 ? st[7] : (r == 3 ? st[8] : (r == 4 ? st[9] : (r == 5 ? st[10] : 0.0)))));
        VRIV = st[11];
    }

    __device__ void flows_of_next_step(double, double, double, double *) const {}

    // x of lane K of this lane's row
    template <int K>
    __device__ __forceinline__ static double bcast(double x)
    {
        double y;
        if constexpr (K == 0)
            asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1" SMART_L_DPP(0) : "=v"(y) : "v"(x));
        else if constexpr (K == 1)
            asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1" SMART_L_DPP(1) : "=v"(y) : "v"(x));
        else if constexpr (K == 2)
            asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1" SMART_L_DPP(2) : "=v"(y) : "v"(x));
        else if constexpr (K == 3)
            asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1" SMART_L_DPP(3) : "=v"(y) : "v"(x));
        else if constexpr (K == 4)
            asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1" SMART_L_DPP(4) : "=v"(y) : "v"(x));
        else if constexpr (K == 5)
            asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1" SMART_L_DPP(5) : "=v"(y) : "v"(x));
        else if constexpr (K == 8)
            asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1" SMART_L_DPP(8) : "=v"(y) : "v"(x));
        else if constexpr (K == 12)
            asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1" SMART_L_DPP(12) : "=v"(y) : "v"(x));
        else {
            static_assert(K == 13, "no such lane role");
            asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1" SMART_L_DPP(13) : "=v"(y) : "v"(x));
        }
        return y;
    }

    __device__ void get_vars(double *v, const double * = nullptr) const
    {
        // (the last step was a quick one iff the last chunk was: `quick`; the guarded form leaves its out[0] in out0)
        v[0] = quick ? Slow::dv<true>(Slow::dv<true>(aeva_last, 1e3, 1e-3) * s.area, s.dt, s.y_dt) : out0;
        v[1] = bcast<kOve>(outq);
        v[2] = bcast<kDra>(outq);
        v[3] = bcast<kInt>(outq);
        v[4] = bcast<kSgw>(outq);
        v[5] = bcast<kDgw>(outq);
        v[6] = out6;
        v[7] = bcast<kOve>(V);
        v[8] = bcast<kDra>(V);
        v[9] = bcast<kInt>(V);
        v[10] = bcast<kSgw>(V);
        v[11] = bcast<kDgw>(V);
        v[12] = bcast<0>(LY);
        v[13] = bcast<1>(LY);
        v[14] = bcast<2>(LY);
        v[15] = bcast<3>(LY);
        v[16] = bcast<4>(LY);
        v[17] = bcast<5>(LY);
        v[18] = VRIV;
    }

    __device__ double excess(double rain_in, double peva_in) const
    {
#pragma clang fp contract(off)
        return rain_in * s.pT - peva_in;
    }

    // ahead of a chunk of (at most) four steps: what LiteralModelT<true>::begin_chunk looks at, over the row's state
    // registers -- and that no layer stands above its capacity (by more than rounding: zcap = z (1 + 2^-40)), which is what
    // keeps s' and the overland share in range for the whole chunk (see step).  The level is the one the chunk's first
    // step computes anyway.
    __device__ __forceinline__ void begin_chunk(const double2 *f4, int n)
    {
#pragma clang fp contract(off)
        const double l = Slow::dv<true>(LY, s.area, s.y_area) * 1e3;
        // (the reservoirs are not looked at: their one quotient V / k is a true division, see step)
        const bool ok = (!is_layer || (Slow::in_range(LY) && l <= zcap)) && Slow::in_range(VRIV);
        bool forcing_ok = true;
        for (int j = 0; j < n; ++j)
            forcing_ok = forcing_ok && Slow::in_range(f4[j].x) && Slow::in_range(f4[j].y);
        quick = s.divisors_fit && forcing_ok && __builtin_amdgcn_ballot_w64(!ok) == 0;
#ifdef SMART_LANES_COUNT
        n_why[0] += !s.divisors_fit;
        n_why[1] += !forcing_ok;
        n_why[2] += __builtin_amdgcn_ballot_w64(is_layer && !Slow::in_range(LY)) != 0;
        n_why[3] += __builtin_amdgcn_ballot_w64(is_layer && !(l <= zcap)) != 0;
        n_why[4] += __builtin_amdgcn_ballot_w64(!Slow::in_range(VRIV)) != 0;
#endif
    }

    // ---- the step, in four asm blocks with hipcc's code between them ---------------------------------------------------
    // Inline asm is opaque to hipcc's hazard recogniser, and gfx950 wants two wait states between a vector instruction
    // that writes a register and a DPP instruction that reads it (src0 of the v_mov_b64_dpp / v_fmac_f64_dpp below; the
    // accumulator is no DPP operand).  Every block is laid out so that at least two instructions of its own stand
    // between the producer of such a register and its first DPP read -- no s_nop anywhere (each would cost a lone
    // wavefront the four cycles of a real instruction); tests/test_lanes_isa.py checks the distance in the built library.
    //
    // (1) levels and excess (:339-355).  t = V_ly / area;  l = t * 1e3;  rain = rain_in * T;  ex = rain - peva_in;
    // tot = ((((l[0] + l[1]) + l[2]) + l[3]) + l[4]) + l[5] over the layers' lanes, in every lane of the row.  Python's
    // sum() starts from 0.0 (:350, and the reference writes 0.0 + l1 itself): 0.0 + l[0] is l[0] for every l[0] but -0.0,
    // which a level never is (states of a quick chunk are +0 or positive: in_range).
    __device__ __forceinline__ void levels_and_excess(double t, double rain_in, double peva_in, double &l, double &tot,
                                                      double &rain, double &ex) const
    {
        asm volatile("v_mul_f64 %[l], %[t], %[k]\n\t"
                     "v_mul_f64 %[rain], %[pT], %[rin]\n\t"
                     "v_add_f64 %[ex], %[rain], -%[pe]\n\t"
                     "v_mov_b64_dpp %[tot], %[l]" SMART_L_DPP(0) "v_fmac_f64_dpp %[tot], %[l], %[one]" SMART_L_DPP(1)
                     "v_fmac_f64_dpp %[tot], %[l], %[one]" SMART_L_DPP(2) "v_fmac_f64_dpp %[tot], %[l], %[one]" SMART_L_DPP(3)
                     "v_fmac_f64_dpp %[tot], %[l], %[one]" SMART_L_DPP(4) "v_fmac_f64_dpp %[tot], %[l], %[one]" SMART_L_DPP(5)
                     : [l] "=&v"(l), [tot] "=&v"(tot), [rain] "=&v"(rain), [ex] "=&v"(ex)
                     : [t] "v"(t), [k] "s"(1e3), [pT] "v"(s.pT), [rin] "s"(rain_in), [pe] "s"(peva_in), [one] "v"(one));
    }
    // (2) the wet side's head (:363-379) and the filling (:367-374).  sp = z - l;  q = tot / Z (reciprocal + correction
    // step);  hp = H q;  e = ex - hp ex (the overland lane keeps ex: its inflow is hp ex);  s1 = S q;  then
    // e = what is left of the excess above this lane's layer (lanes that are no layer: below all six)
#define SMART_L_FILL(k, mk)                                                                                            \
    "v_fmac_f64_dpp %[e], %[sp], %[" mk "]" SMART_L_DPP(k) "v_max_f64 %[e], %[e], 0\n\t"
    __device__ __forceinline__ void wet_head_and_fill(double l, double tot, double ex, double &sp, double &hp, double &e,
                                                      double &s1) const
    {
        double q, r;
        asm volatile("v_add_f64 %[sp], %[z], -%[l]\n\t"
                     "v_mul_f64 %[q], %[tot], %[yz]\n\t"
                     "v_fma_f64 %[r], -%[pZ], %[q], %[tot]\n\t"
                     "v_fmac_f64_e32 %[q], %[r], %[yz]\n\t"
                     "v_mul_f64 %[hp], %[pH], %[q]\n\t"
                     "v_mul_f64 %[r], %[hp], %[ex]\n\t"
                     "v_fma_f64 %[e], -%[r], %[wof], %[ex]\n\t"
                     "v_mul_f64 %[s1], %[pS], %[q]\n\t" SMART_L_FILL(0, "m0") SMART_L_FILL(1, "m1") SMART_L_FILL(2, "m2")
                         SMART_L_FILL(3, "m3") SMART_L_FILL(4, "m4") SMART_L_FILL(5, "m5")
                     : [sp] "=&v"(sp), [q] "=&v"(q), [r] "=&v"(r), [hp] "=&v"(hp), [e] "=&v"(e), [s1] "=&v"(s1)
                     : [l] "v"(l), [tot] "v"(tot), [ex] "v"(ex), [z] "v"(z), [yz] "v"(s.y_z), [pZ] "v"(s.pZ), [pH] "v"(s.pH),
                       [pS] "v"(s.pS), [wof] "v"(w_of), [m0] "v"(m[0]), [m1] "v"(m[1]), [m2] "v"(m[2]), [m3] "v"(m[3]),
                       [m4] "v"(m[4]), [m5] "v"(m[5]));
    }
    // (3) the three leak passes (:381-399): x = l f; l -= x for every layer at once, and the pass's total as
    // (((((acc + x[0]) + x[1]) + x[2]) + x[3]) + x[4]) + x[5] -- bottom layer first in the deep pass -- added into the
    // lanes of ONE bank of X (the reservoir the total belongs to sits there: interflow bank 2, shallow groundwater bank 0,
    // deep groundwater bank 1); the other banks keep theirs.  A pass's products are formed while the chain of the pass
    // before it runs.
#define SMART_L_SUM(x, b)                                                                                              \
    "v_fmac_f64_dpp %[X], %[" x "], %[one]" SMART_L_DPPB(0, b) "v_fmac_f64_dpp %[X], %[" x "], %[one]" SMART_L_DPPB(1, b)  \
    "v_fmac_f64_dpp %[X], %[" x "], %[one]" SMART_L_DPPB(2, b) "v_fmac_f64_dpp %[X], %[" x "], %[one]" SMART_L_DPPB(3, b)  \
    "v_fmac_f64_dpp %[X], %[" x "], %[one]" SMART_L_DPPB(4, b) "v_fmac_f64_dpp %[X], %[" x "], %[one]" SMART_L_DPPB(5, b)
    __device__ __forceinline__ void leaks(double &X, double &l, double f1, double f2, double f3) const
    {
        double x1, x2, x3;
        asm volatile("v_mul_f64 %[x1], %[l], %[f1]\n\t"
                     "v_add_f64 %[l], %[l], -%[x1]\n\t"
                     "v_mul_f64 %[x2], %[l], %[f2]\n\t" SMART_L_SUM("x1", 0x4)
                     "v_add_f64 %[l], %[l], -%[x2]\n\t"
                     "v_mul_f64 %[x3], %[l], %[f3]\n\t" SMART_L_SUM("x2", 0x1)
                     "v_add_f64 %[l], %[l], -%[x3]\n\t"
                     "v_fmac_f64_dpp %[X], %[x3], %[one]" SMART_L_DPPB(5, 0x2) "v_fmac_f64_dpp %[X], %[x3], %[one]" SMART_L_DPPB(4, 0x2)
                     "v_fmac_f64_dpp %[X], %[x3], %[one]" SMART_L_DPPB(3, 0x2) "v_fmac_f64_dpp %[X], %[x3], %[one]" SMART_L_DPPB(2, 0x2)
                     "v_fmac_f64_dpp %[X], %[x3], %[one]" SMART_L_DPPB(1, 0x2) "v_fmac_f64_dpp %[X], %[x3], %[one]" SMART_L_DPPB(0, 0x2)
                     : [X] "+v"(X), [l] "+v"(l), [x1] "=&v"(x1), [x2] "=&v"(x2), [x3] "=&v"(x3)
                     : [f1] "v"(f1), [f2] "v"(f2), [f3] "v"(f3), [one] "v"(one));
    }
    // (2') the dry side (:400-419): d = -ex is the demand; the demand that arrives at this lane's layer, C (d - l) handed
    // down while d > l; what the layer gives is min(l, d) either way; aeva = rain + the six takes in the layers' order;
    // nothing flows into the reservoirs (X = 0)
#define SMART_L_EVAP(k, mk, ck)                                                                                        \
    "v_fmac_f64_dpp %[d], %[l], %[" mk "]" SMART_L_DPP(k) "v_max_f64 %[d], %[d], 0\n\t"                                 \
                                                          "v_mul_f64 %[d], %[" ck "], %[d]\n\t"
    __device__ __forceinline__ void evaporation(double &aeva, double &l, double &X, double ex) const
    {
        double d;
        asm volatile("v_mul_f64 %[d], %[ex], -1.0\n\t" SMART_L_EVAP(0, "m0", "c0") SMART_L_EVAP(1, "m1", "c1")
                         SMART_L_EVAP(2, "m2", "c2") SMART_L_EVAP(3, "m3", "c3") SMART_L_EVAP(4, "m4", "c4")
                             SMART_L_EVAP(5, "m5", "c5")
                     "v_min_f64 %[d], %[l], %[d]\n\t"
                     "v_add_f64 %[l], %[l], -%[d]\n\t"
                     "v_mov_b64 %[X], 0\n\t"
                     "v_fmac_f64_dpp %[a], %[d], %[one]" SMART_L_DPP(0) "v_fmac_f64_dpp %[a], %[d], %[one]" SMART_L_DPP(1)
                     "v_fmac_f64_dpp %[a], %[d], %[one]" SMART_L_DPP(2) "v_fmac_f64_dpp %[a], %[d], %[one]" SMART_L_DPP(3)
                     "v_fmac_f64_dpp %[a], %[d], %[one]" SMART_L_DPP(4) "v_fmac_f64_dpp %[a], %[d], %[one]" SMART_L_DPP(5)
                     : [d] "=&v"(d), [l] "+v"(l), [a] "+v"(aeva), [X] "=&v"(X)
                     : [ex] "v"(ex), [one] "v"(one), [m0] "v"(m[0]), [m1] "v"(m[1]), [m2] "v"(m[2]), [m3] "v"(m[3]),
                       [m4] "v"(m[4]), [m5] "v"(m[5]), [c0] "v"(cm[0]), [c1] "v"(cm[1]), [c2] "v"(cm[2]), [c3] "v"(cm[3]),
                       [c4] "v"(cm[4]), [c5] "v"(cm[5]));
    }
    // (4) the five reservoirs at once (:427-450), q = V / k given:  V += (X / 1e3) area - q dt;  "if V < 0: V = 0" as a
    // maximum (V is never -0.0 or a NaN on this path); then the river's inflow (:254) out[1] + out[2] + out[3] + out[4] +
    // out[5] and its groundwater part out[4] + out[5], summed over the reservoirs' lanes
    __device__ __forceinline__ void route_and_sum(double &Vr, double X, double q, double &qin, double &qgw) const
    {
        double t, r;
        asm volatile("v_mul_f64 %[t], %[X], %[km]\n\t"
                     "v_fma_f64 %[r], -%[k], %[t], %[X]\n\t"
                     "v_fmac_f64_e32 %[t], %[km], %[r]\n\t"
                     "v_mul_f64 %[t], %[t], %[area]\n\t"
                     "v_mul_f64 %[r], %[q], %[dt]\n\t"
                     "v_add_f64 %[t], %[t], -%[r]\n\t"
                     "v_add_f64 %[V], %[V], %[t]\n\t"
                     "v_max_f64 %[V], %[V], 0\n\t"
                     "v_mov_b64_dpp %[qi], %[q]" SMART_L_DPP(12) "v_mov_b64_dpp %[qg], %[q]" SMART_L_DPP(0)
                     "v_fmac_f64_dpp %[qi], %[q], %[one]" SMART_L_DPP(13) "v_fmac_f64_dpp %[qg], %[q], %[one]" SMART_L_DPP(4)
                     "v_fmac_f64_dpp %[qi], %[q], %[one]" SMART_L_DPP(8) "v_fmac_f64_dpp %[qi], %[q], %[one]" SMART_L_DPP(0)
                     "v_fmac_f64_dpp %[qi], %[q], %[one]" SMART_L_DPP(4)
                     : [V] "+v"(Vr), [t] "=&v"(t), [r] "=&v"(r), [qi] "=&v"(qin), [qg] "=&v"(qgw)
                     : [X] "v"(X), [q] "v"(q), [k] "s"(1e3), [km] "s"(1e-3), [area] "v"(s.area), [dt] "v"(s.dt),
                       [one] "v"(one));
    }
    // lane i <- lane i + 8 of the row (the deep pass's powers come down from the lanes they were made in; lanes 8 to 15
    // keep whatever the registers held: they are no layer)
    __device__ __forceinline__ static double down8(double x)
    {
        const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), 0x108, 0xf, 0xf, false);
        const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), 0x108, 0xf, 0xf, false);
        return __hiloint2double(hi, lo);
    }

    __device__ void step(double rain_in, double peva_in, double /*ex*/, double &acc, double &num, double &den)
    {
#pragma clang fp contract(off)
        if (__builtin_expect(!quick, 0)) {
#ifdef SMART_LANES_COUNT
            ++n_slow;
#endif
            slow_step(rain_in, peva_in, acc, num, den);
            return;
        }
        const double area = s.area, dt = s.dt;
        double l, tot, rain, ex0;
        {
            // V_ly / area by reciprocal + correction step, short of the last multiplication (block 1 has it)
            const double q0 = LY * s.y_area;
            const double r = __builtin_fma(-area, q0, LY);
            levels_and_excess(__builtin_fma(r, s.y_area, q0), rain_in, peva_in, l, tot, rain, ex0);
        }
        double aeva, X;
        if (ex0 >= 0.0) { // :359.  Rows on the wet side (usually all four or none: the forcing is the wave's)
            aeva = peva_in; // 0.0 + peva_in: the forcing of a quick chunk is +0 or a positive normal number
            // What LiteralModelT<true> asks of every wet step before it drops the leak guards -- 0 <= s' <= 0.75 and an excess
            // that the overland share leaves >= 0 -- holds for every step of a quick chunk: its layers start within
            // capacity (begin_chunk) and filling, leaks and evaporation keep them there, so tot / Z <= 1 + 2^-39, and
            // H <= 0.99, 0 <= S <= 0.74 are conditions of the run (setup).
            double sp, hp, e, s1;
            wet_head_and_fill(l, tot, ex0, sp, hp, e, s1);
            l = e <= sp ? l + e : z; // :367-374
            const double a = __builtin_fma(hp, sx, kx);
            X = a * e; // ove: hp * ex0; dra: pD * ex; int: (1 - pD) * ex; sgw, dgw: 0
            double P = s1; // s'^i as the product chain, each lane as far as its layer wants it
#pragma unroll
            for (int k = 0; k < 5; ++k)
                P = P * __builtin_fma(s1, dk[k], ek[k]);
            const double f2 = Slow::dv<true>(s1, nby, by); // s' / (layer + 1)
            leaks(X, l, P, f2, down8(P));
        } else { // :400
            aeva = rain; // 0.0 + rain: rain = rain_in * T is +0 or positive (T >= 0 is one of the run's conditions)
            evaporation(aeva, l, X, ex0);
        }
        aeva_last = aeva; // (:424: the evaporation flow of the step is worked out when somebody asks for it, get_vars)
        // V / k as a TRUE division: a reservoir that is rarely fed (the drain: saturation excess only) decays geometrically
        // below the range the correction step is proven for, stays there for hundreds of steps, and in this form the five
        // quotients are one instruction sequence anyway (ten instructions instead of three)
        outq = V / kres;
        route_and_sum(V, X, outq, q_in, q_gw);
        LY = Slow::dv<true>(l, 1e3, 1e-3) * area; // :456-457
        const double q = Slow::river_q<true>(dt, q_in, s.rk, VRIV, s.y_rk, s.y_dt);
        out6 = q;
        q_out = q;
        acc += q;
        num += q_gw;
        den += q_in;
    }

    // the guarded form: LiteralModel's own step on the row's states (literal_guarded_step above), every lane of the row
    // the same computation, each taking its own role's results back
    __device__ __forceinline__ void slow_step(double rain_in, double peva_in, double &acc, double &num, double &den)
    {
        double io[35];
        io[0] = s.pT, io[1] = s.pC, io[2] = s.pH, io[3] = s.pD, io[4] = s.pS, io[5] = s.pZ;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            io[6 + i] = raw_k[i];
        io[10] = s.area, io[11] = s.dt;
        io[12] = bcast<kOve>(V);
        io[13] = bcast<kDra>(V);
        io[14] = bcast<kInt>(V);
        io[15] = bcast<kSgw>(V);
        io[16] = bcast<kDgw>(V);
        io[17] = bcast<0>(LY);
        io[18] = bcast<1>(LY);
        io[19] = bcast<2>(LY);
        io[20] = bcast<3>(LY);
        io[21] = bcast<4>(LY);
        io[22] = bcast<5>(LY);
        io[23] = VRIV;
        io[24] = rain_in, io[25] = peva_in;
        literal_guarded_step(io);
        const int r = role;
        V = r == kOve ? io[12] : (r == kDra ? io[13] : (r == kInt ? io[14] : (r == kSgw ? io[15] : (r == kDgw ? io[16] : 0.0))));
        LY = r == 0 ? io[17] : (r == 1 ? io[18] : (r == 2 ? io[19] : (r == 3 ? io[20] : (r == 4 ? io[21] : (r == 5 ? io[22] : 0.0)))));
        VRIV = io[23];
        out0 = io[26];
        outq = r == kOve ? io[27] : (r == kDra ? io[28] : (r == kInt ? io[29] : (r == kSgw ? io[30] : (r == kDgw ? io[31] : 0.0))));
        out6 = io[32];
        q_out = io[32];
        q_in = io[33];
        q_gw = io[34];
        acc += q_out;
        num += q_gw;
        den += q_in;
    }
};

} // namespace smart
