// smart_fast_arms.h -- the step loop of sub-daily forcing at the instruction level: the TEXT of the gfx950 code, as
// string macros that FastModel::step_arms / run_chunks (smart_fast_model.h) put into `asm` statements.
//
// step_lazy() (smart_fast_model.h) is ONE body for every kind of step, built around a predicated region; hipcc pays
// for every other shape with register copies at the joins.  The forcing of a step is wave-uniform,
// so what a step needs is known on the scalar unit before any vector work:
//   rain == 0, peva > 0   DRY   every lane is on the dry side (ex = -peva whatever T is): route, add the demand to
//                               `pend`, drain the reservoirs -- 9 vector instructions, no compare, no EXEC change;
//   rain == 0, peva == 0  CALM  every lane is wet with zero excess: route, the three leak passes, reservoirs -- 51
//                               instructions, no EXEC change, no filling;
//   rain > 0              RAIN  the general step: excess per lane, wet lanes under EXEC, filling, leaks -- 84, of
//                               which 17 (the filling below the top layer) are skipped when the top layer takes the
//                               excess of every wet lane (SMART_RAIN_FILL_EXIT: 59 % of the rainy steps).
// The deferred evaporation cascade (FastModel::flush_pending) is due in the calm and the rain arm when a wet lane has
// a demand pending -- read from `pend` itself (lanes with something pending are exactly the lanes with pend > 0; the
// lane mask step_lazy() carries says the same, but an SGPR pair that lives across `asm` statements is taken for
// divergent by hipcc's uniformity analysis and refused).  It is due on one step in twelve and sits OUT OF LINE, in
// the shadow of an unconditional branch, so that the usual step falls through.
//
// Operands pin every state to its register for good: arms meet without a copy.  The arithmetic is that of
// step_lazy(), operation for operation (same operand order, same contractions): outputs are bit-identical
// (tools/debug/steps_bits.py against a -DSMART_STEP_ARMS=0 build; profiles/r03_steps_bits.txt).
// The shortcuts of the dry and the calm arm hold for forcing whose values are all finite and >= +0
// (smart_forcing_scan) and waves with no layer above capacity (zero_ok): QUICK.  Any other wave takes the rain arm
// for every step, which is step_lazy() without the calm shortcut.
//
// Two shapes:
//   SMART_A_STEP      one step: dispatch, three arms (report gaps that are not a multiple of the chunk, tails);
//   SMART_A_CHUNK     the four steps of a chunk (one 64-byte line of forcing), THREADED: the three arms are laid out
//                     as three lanes of four steps each, and every arm ends with the dispatch of the NEXT step, whose
//                     fall-through is the same lane -- a step of the kind of its predecessor (two in three) costs no
//                     taken branch at all, any other exactly one.  A lone wave pays ~60 cycles for a taken branch
//                     (profiles/r01_microbench_valu_salu_branch.txt), a dry step is 36 cycles of vector work.
//
// Manual hazards (gfx950; inline asm is opaque to hipcc's hazard recogniser): no vector instruction reads an SGPR
// within two instructions of the v_cmp that wrote it (masks are read by scalar instructions, ten instructions
// later); v_cmp -> s_cbranch_vccz/nz is interlocked; scalar loads are waited for with s_waitcnt lgkmcnt(0) before
// their destinations are read.
#pragma once

// ---- pieces ---------------------------------------------------------------------------------------------------
// onto an 8-byte boundary (at most one s_nop): fp64 instructions are 64-bit encodings, and a run of them that starts at
// 4 mod 8 straddles a 32-byte fetch line every fourth instruction (SMART_A_WET_INTERVAL has the story and the numbers)
#define SMART_A_ALIGN8 ".p2align 3\n\t"
// ... onto 4 mod 8: what the RAIN arm wants for its first instruction -- a 4-byte v_mov_b64, behind which its first ten
// instructions then lie on the boundary; the three 4-byte v_mov_b64 and four scalar instructions that follow leave the
// seven of the filling's head at 4 mod 8, and the fill exit's s_cbranch puts the arm's other 59 back on the boundary
#define SMART_A_ALIGN8_ODD ".p2align 3\n\ts_nop 0\n\t"
#define SMART_A_ROUTE                                                                                                  \
    "v_mul_f64 %[t0], %[cg], %[yg]\n\t"                                                                                \
    "v_fma_f64 %[t0], %[cf], %[yf], %[t0]\n\t"                                                                         \
    "v_fma_f64 %[t0], %[cs], %[ys], %[t0]\n\t"                                                                         \
    "v_add_f64 %[acc], %[acc], %[riv]\n\t"                                                                             \
    "v_fma_f64 %[riv], %[riv], %[oma], %[t0]\n\t"
// ... for reports that want the river's outflow of ONE step -- raw reports (structure.py:192-195: the last step of each
// interval) and a report every step (gap 1) -- instead of the interval's sum: the same five instructions, but `acc`
// ends up holding the outflow of the step just taken (the river's state ahead of its update) and the routing temporary
// is kept in registers of its own: qi = dt/rk * (the step's inflow to the river, sum of the five catchment outflows),
// qg = dt/rk * (its groundwater part) -- the two sums behind the groundwater ratio of raw reports (:194-195; the
// common factor dt/rk cancels in the ratio).  The river update rounds exactly as in SMART_A_ROUTE.  (The move in its
// 64-bit encoding, _e64: five 8-byte instructions like SMART_A_ROUTE's, the parity of what follows is the same.)
#define SMART_A_ROUTE_LAST                                                                                             \
    "v_mul_f64 %[qg], %[cg], %[yg]\n\t"                                                                                \
    "v_fma_f64 %[qi], %[cf], %[yf], %[qg]\n\t"                                                                         \
    "v_fma_f64 %[qi], %[cs], %[ys], %[qi]\n\t"                                                                         \
    "v_mov_b64_e64 %[acc], %[riv]\n\t"                                                                                 \
    "v_fma_f64 %[riv], %[riv], %[oma], %[qi]\n\t"
// deferred evaporation cascade over the active lanes (flush_pending): t = d - l; l = max(-t, 0); d = max(C t, 0)
#define SMART_A_CASCADE                                                                                                \
    "v_add_f64 %[l0], %[pend], -%[l0]\n\t"                                                                             \
    "v_mul_f64 %[t0], %[pc], %[l0]\n\t"                                                                                \
    "v_max_f64 %[t0], %[t0], 0\n\t"                                                                                    \
    "v_cmp_lt_f64 vcc, 0, %[t0]\n\t"                                                                                   \
    "s_cbranch_vccz 2f\n\t"                                                                                            \
    "v_add_f64 %[l1], %[t0], -%[l1]\n\t"                                                                               \
    "v_mul_f64 %[t0], %[pc], %[l1]\n\t"                                                                                \
    "v_max_f64 %[t0], %[t0], 0\n\t"                                                                                    \
    "v_cmp_lt_f64 vcc, 0, %[t0]\n\t"                                                                                   \
    "s_cbranch_vccz 1f\n\t"                                                                                            \
    "v_add_f64 %[t0], %[t0], -%[l2]\n\t"                                                                               \
    "v_max_f64 %[l2], -%[t0], 0\n\t"                                                                                   \
    "v_mul_f64 %[t0], %[pc], %[t0]\n\t"                                                                                \
    "v_max_f64 %[t0], %[t0], 0\n\t"                                                                                    \
    "v_add_f64 %[t0], %[t0], -%[l3]\n\t"                                                                               \
    "v_max_f64 %[l3], -%[t0], 0\n\t"                                                                                   \
    "v_mul_f64 %[t0], %[pc], %[t0]\n\t"                                                                                \
    "v_max_f64 %[t0], %[t0], 0\n\t"                                                                                    \
    "v_add_f64 %[t0], %[t0], -%[l4]\n\t"                                                                               \
    "v_max_f64 %[l4], -%[t0], 0\n\t"                                                                                   \
    "v_mul_f64 %[t0], %[pc], %[t0]\n\t"                                                                                \
    "v_max_f64 %[t0], %[t0], 0\n\t"                                                                                    \
    "v_add_f64 %[l5], %[t0], -%[l5]\n\t"                                                                               \
    "v_max_f64 %[l5], -%[l5], 0\n\t"                                                                                   \
    "1:\n\t"                                                                                                           \
    "v_max_f64 %[l1], -%[l1], 0\n\t"                                                                                   \
    "2:\n\t"                                                                                                           \
    "v_max_f64 %[l0], -%[l0], 0\n\t"                                                                                   \
    "v_mov_b64 %[pend], 0\n\t"                                                                                         \
    "v_add_f64 %[tot], %[l0], %[l1]\n\t"                                                                               \
    "v_add_f64 %[t0], %[l2], %[l3]\n\t"                                                                                \
    "v_add_f64 %[tot], %[tot], %[t0]\n\t"                                                                              \
    "v_add_f64 %[t0], %[l4], %[l5]\n\t"                                                                                \
    "v_add_f64 %[tot], %[tot], %[t0]\n\t"
// top-down filling (structure.py:363-377): ex_in in xf, the saturation excess ends in t1
#define SMART_A_FILL1(l, src)                                                                                          \
    "v_add_f64 %[t1], %[" l "], %[" src "]\n\t"                                                                        \
    "v_min_f64 %[" l "], %[t1], %[z]\n\t"                                                                              \
    "v_add_f64 %[t1], %[t1], -%[" l "]\n\t"
#define SMART_A_FILL_REST                                                                                              \
    "v_fma_f64 %[xf], -%[eh], %[tot], %[ex]\n\t" SMART_A_FILL1("l0", "xf") SMART_A_FILL1("l1", "t1")                   \
        SMART_A_FILL1("l2", "t1") SMART_A_FILL1("l3", "t1") SMART_A_FILL1("l4", "t1") SMART_A_FILL1("l5", "t1")        \
            "v_mul_f64 %[xs], %[eh], %[tot]\n\t"                                                                       \
            "v_fma_f64 %[xf], -%[pd], %[t1], %[xf]\n\t"                                                                \
            "v_fma_f64 %[xs], %[pd], %[t1], %[xs]\n\t"
#define SMART_A_FILL(drain) "v_mul_f64 %[eh], %[hz], %[ex]\n\t" SMART_A_FILL_REST drain
// ... with a way out behind the top layer: when no lane has excess left there (59 % of the rainy steps of the
// flat-forcing workload) the other five layers see  t = l + 0; l = min(t, z); t - l = 0  -- the identity as long as
// no layer is above its capacity (QUICK waves only) -- and the saturation excess is zero (v_cmp_nle: a NaN goes on)
#ifndef SMART_RAIN_FILL_EXIT
#define SMART_RAIN_FILL_EXIT 1
#endif
#if SMART_RAIN_FILL_EXIT
#define SMART_A_FILL_QUICK(drain)                                                                                      \
    "v_mul_f64 %[eh], %[hz], %[ex]\n\t"                                                                                \
    "v_fma_f64 %[xf], -%[eh], %[tot], %[ex]\n\t" SMART_A_FILL1("l0", "xf") "v_mul_f64 %[xs], %[eh], %[tot]\n\t"        \
                                                                           "v_cmp_nle_f64 vcc, %[t1], 0\n\t"           \
                                                                           "s_cbranch_vccz 6f\n\t" SMART_A_FILL1(      \
                                                                               "l1", "t1") SMART_A_FILL1("l2", "t1")   \
        SMART_A_FILL1("l3", "t1") SMART_A_FILL1("l4", "t1") SMART_A_FILL1("l5", "t1")                                  \
            "v_fma_f64 %[xf], -%[pd], %[t1], %[xf]\n\t"                                                                \
            "v_fma_f64 %[xs], %[pd], %[t1], %[xs]\n\t" drain "6:\n\t"
#else
#define SMART_A_FILL_QUICK(drain) SMART_A_FILL(drain)
#endif
// s', s'^2 and s'^3 live in the registers of three temporaries that are dead by the time the leaks start (the excess,
// e_h and the routing / cascade temporary): 6 VGPRs less
#define SMART_S1 "ex"
#define SMART_P2 "eh"
#define SMART_P3 "t0"
#define SMART_A_LEAK(l, p) "v_fma_f64 %[" l "], -%[" l "], %[" p "], %[" l "]\n\t"
#define SMART_A_LSUM(dst)                                                                                              \
    "v_add_f64 %[" dst "], %[l1], %[l0]\n\t"                                                                           \
    "v_add_f64 %[t1], %[l3], %[l2]\n\t"                                                                                \
    "v_add_f64 %[" dst "], %[t1], %[" dst "]\n\t"                                                                      \
    "v_add_f64 %[t1], %[l5], %[l4]\n\t"                                                                                \
    "v_add_f64 %[" dst "], %[t1], %[" dst "]\n\t"
// the three leak passes (structure.py:381-399); `deep`: the SPLIT models' sum of what the third pass takes
#define SMART_A_LEAKS_(S1, P2, P3, deep)                                                                               \
    "v_mul_f64 %[" S1 "], %[sz], %[tot]\n\t"                                                                           \
    "v_mul_f64 %[" P2 "], %[" S1 "], %[" S1 "]\n\t"                                                                    \
    "v_mul_f64 %[" P3 "], %[" S1 "], %[" P2 "]\n\t"                                                                    \
    "v_mul_f64 %[p4], %[" P2 "], %[" P2 "]\n\t"                                                                        \
    "v_mul_f64 %[p5], %[" S1 "], %[p4]\n\t"                                                                            \
    "v_mul_f64 %[p6], %[" P3 "], %[" P3 "]\n\t" SMART_A_LEAK("l0", S1) SMART_A_LEAK("l1", P2)                          \
        SMART_A_LEAK("l2", P3) SMART_A_LEAK("l3", "p4") SMART_A_LEAK("l4", "p5") SMART_A_LEAK("l5", "p6")              \
            SMART_A_LSUM("ai") SMART_A_LEAK("l0", S1) "v_mul_f64 %[t1], %[" S1 "], -0.5\n\t"                           \
                                                            "v_fma_f64 %[l1], %[l1], %[t1], %[l1]\n\t"                 \
                                                            "v_mul_f64 %[t1], %[" S1 "], %[k3]\n\t"                    \
                                                            "v_fma_f64 %[l2], %[l2], %[t1], %[l2]\n\t"                 \
                                                            "v_ldexp_f64 %[t1], -%[" S1 "], -2\n\t"                    \
                                                            "v_fma_f64 %[l3], %[l3], %[t1], %[l3]\n\t"                 \
                                                            "v_mul_f64 %[t1], %[" S1 "], %[k5]\n\t"                    \
                                                            "v_fma_f64 %[l4], %[l4], %[t1], %[l4]\n\t"                 \
                                                            "v_mul_f64 %[t1], %[" S1 "], %[k6]\n\t"                    \
                                                            "v_fma_f64 %[l5], %[l5], %[t1], %[l5]\n\t" deep            \
                SMART_A_LEAK("l0", "p6") SMART_A_LEAK("l1", "p5") SMART_A_LEAK("l2", "p4")                             \
                    SMART_A_LEAK("l3", P3) SMART_A_LEAK("l4", P2) SMART_A_LEAK("l5", S1)
#define SMART_A_LEAKS(deep) SMART_A_LEAKS_(SMART_S1, SMART_P2, SMART_P3, deep)
#define SMART_A_DEEP                                                                                                   \
    "v_mul_f64 %[dp], %[l0], %[p6]\n\t"                                                                                \
    "v_fma_f64 %[dp], %[l1], %[p5], %[dp]\n\t"                                                                         \
    "v_fma_f64 %[dp], %[l2], %[p4], %[dp]\n\t"                                                                         \
    "v_fma_f64 %[dp], %[l3], %[" SMART_P3 "], %[dp]\n\t"                                                               \
    "v_fma_f64 %[dp], %[l4], %[" SMART_P2 "], %[dp]\n\t"                                                               \
    "v_fma_f64 %[dp], %[l5], %[" SMART_S1 "], %[dp]\n\t"
#define SMART_A_TOT_XG SMART_A_LSUM("tot") "v_add_f64 %[xg], %[ai], -%[tot]\n\t"

// ---- the three arms.  route: SMART_A_ROUTE (interval sums) or SMART_A_ROUTE_LAST; pe / rn: the TEXT of the SGPR pairs
// that hold the step's forcing (an asm operand, "%[pe0]", or a physical register, "s[38:39]"); casc: the hook of the deferred cascade (SMART_A_CASC_*); deep / split / zeros / drain:
// what the SPLIT models add ("" otherwise)
#define SMART_A_DRY(route, pe, split)                                                                                  \
    route "v_add_f64 %[pend], %[pend], " pe "\n\t"                                                          \
                  "v_fma_f64 %[ys], %[ys], %[ds], 0\n\t"                                                               \
                  "v_fma_f64 %[yf], %[yf], %[df], 0\n\t"                                                               \
                  "v_fma_f64 %[yg], %[yg], %[dg], 0\n\t" split
#define SMART_A_DRY_SPLIT                                                                                              \
    "v_fma_f64 %[yd], %[yd], %[ds], 0\n\t"                                                                             \
    "v_fma_f64 %[ydg], %[ydg], %[dg], 0\n\t"
#define SMART_A_CALM_BODY(route, casc, deep, split)                                                                    \
    route casc SMART_A_LEAKS(deep) "v_add_f64 %[xf], %[tot], -%[ai]\n\t"                                               \
        SMART_A_TOT_XG "v_fma_f64 %[ys], %[ys], %[ds], 0\n\t"                                                          \
                       "v_fma_f64 %[yf], %[yf], %[df], %[xf]\n\t"                                                      \
                       "v_fma_f64 %[yg], %[yg], %[dg], %[xg]\n\t"                                                      \
                       "v_add_f64 %[xgs], %[xgs], %[xg]\n\t" split
#define SMART_A_CALM(route, casc, deep, split) "v_cmp_lt_f64 vcc, 0, %[pend]\n\t" SMART_A_CALM_BODY(route, casc, deep, split)
#define SMART_A_CALM_SPLIT                                                                                             \
    "v_fma_f64 %[yd], %[yd], %[ds], 0\n\t"                                                                             \
    "v_fma_f64 %[ydg], %[ydg], %[dg], %[dp]\n\t"
// (pendcmp: the test for a pending demand, casc: what acts on it -- the hook of the cascade behind  tmp &= wet lanes;
// the pair blocks, which know the step before, leave both out or put the cascade itself there)
#define SMART_A_RAIN_X(route, rn, pe, pendcmp, casc, zeros, fill, deep, split)                                         \
    "v_mov_b64 %[t1], " pe "\n\t"                                                                                      \
    "v_fma_f64 %[ex], " rn ", %[pt], -%[t1]\n\t"                                                                       \
    "v_cmp_le_f64 %[wm], 0, %[ex]\n\t" pendcmp route "v_max_f64 %[t1], -%[ex], 0\n\t"                                  \
                                                     "v_add_f64 %[pend], %[pend], %[t1]\n\t"                           \
                                                     "v_mov_b64 %[xs], 0\n\t"                                          \
                                                     "v_mov_b64 %[xf], 0\n\t"                                          \
                                                     "v_mov_b64 %[xg], 0\n\t" zeros                                    \
                                                     "s_and_saveexec_b64 %[sv], %[wm]\n\t"                             \
                                                     "s_cbranch_execz 8f\n\t" casc fill                                \
        SMART_A_LEAKS(deep) "v_add_f64 %[t1], %[tot], -%[ai]\n\t"                                                      \
                            "v_add_f64 %[xf], %[xf], %[t1]\n\t" SMART_A_TOT_XG "8:\n\t"                                \
                            "s_or_b64 exec, exec, %[sv]\n\t"                                                           \
                            "v_fma_f64 %[ys], %[ys], %[ds], %[xs]\n\t"                                                 \
                            "v_fma_f64 %[yf], %[yf], %[df], %[xf]\n\t"                                                 \
                            "v_fma_f64 %[yg], %[yg], %[dg], %[xg]\n\t"                                                 \
                            "v_add_f64 %[xgs], %[xgs], %[xg]\n\t" split
#define SMART_A_RAIN(route, rn, pe, casc, zeros, fill, deep, split)                                                    \
    SMART_A_RAIN_X(route, rn, pe, "v_cmp_lt_f64 %[tmp], 0, %[pend]\n\t", "s_and_b64 %[tmp], %[tmp], %[wm]\n\t" casc,    \
                   zeros, fill, deep, split)
#define SMART_A_RAIN_ZEROS_SPLIT "v_mov_b64 %[xd], 0\n\tv_mov_b64 %[dp], 0\n\t"
#define SMART_A_RAIN_DRAIN_SPLIT "v_mul_f64 %[xd], %[pd], %[t1]\n\t"
#define SMART_A_RAIN_SPLIT                                                                                             \
    "v_fma_f64 %[yd], %[yd], %[ds], %[xd]\n\t"                                                                         \
    "v_fma_f64 %[ydg], %[ydg], %[dg], %[dp]\n\t"
// the cascade hook of an arm and its out-of-line block; `id`: digits that make the two labels unique in the asm.
// SMART_ARM_OOL 0 keeps the cascade in line, skipped by a taken branch when it is not due (A/B: tools/gpu_r03_*.sh)
#ifndef SMART_ARM_OOL
#define SMART_ARM_OOL 1
#endif
#if SMART_ARM_OOL
#define SMART_A_CASC_CALM(id) "s_cbranch_vccnz 3" id "0f\n\t3" id "1:\n\t"
#define SMART_A_CASC_CALM_OOL(id) "3" id "0:\n\t" SMART_A_CASCADE "s_branch 3" id "1b\n\t"
#define SMART_A_CASC_RAIN(id) "s_cbranch_scc1 4" id "0f\n\t4" id "1:\n\t"
#define SMART_A_CASC_RAIN_OOL(id) "4" id "0:\n\t" SMART_A_CASCADE "s_branch 4" id "1b\n\t"
#else
#define SMART_A_CASC_CALM(id) "s_cbranch_vccz 3" id "1f\n\t" SMART_A_CASCADE "3" id "1:\n\t"
#define SMART_A_CASC_CALM_OOL(id) ""
#define SMART_A_CASC_RAIN(id) "s_cbranch_scc0 4" id "1f\n\t" SMART_A_CASCADE "4" id "1:\n\t"
#define SMART_A_CASC_RAIN_OOL(id) ""
#endif

// ---- one step ---------------------------------------------------------------------------------------------------
// dispatch, calm arm (entered by falling through), rain arm, dry arm (left by falling through): one taken branch per
// calm or dry step, two per rain step
#define SMART_A_STEP(route, deep, calm_split, zeros, drain, rain_split, dry_split)                                     \
    SMART_A_ALIGN8 "s_cmp_eq_u64 %[rn0], 0\n\t"                                                                                       \
    "s_cbranch_scc0 5f\n\t"                                                                                            \
    "s_cmp_eq_u64 %[pe0], 0\n\t"                                                                                       \
    "s_cbranch_scc0 7f\n\t" SMART_A_CALM(route, SMART_A_CASC_CALM("9"), deep, calm_split) "s_branch 9f\n\t"            \
        SMART_A_CASC_CALM_OOL("9") SMART_A_ALIGN8_ODD "5:\n\t" SMART_A_RAIN(route, "%[rn0]", "%[pe0]", SMART_A_CASC_RAIN("9"), \
                                                                            zeros, SMART_A_FILL_QUICK(drain), deep,    \
                                                                            rain_split)                                \
            "s_branch 9f\n\t" SMART_A_CASC_RAIN_OOL("9") SMART_A_ALIGN8                                                \
            "7:\n\t" SMART_A_DRY(route, "%[pe0]", dry_split) "9:\n\t"
// the rain arm alone (waves that may not take the shortcuts)
#define SMART_A_STEP_RAIN(route, deep, zeros, drain, rain_split)                                                       \
    SMART_A_ALIGN8 SMART_A_RAIN(route, "%[rn0]", "%[pe0]", SMART_A_CASC_RAIN("9"), zeros, SMART_A_FILL(drain), deep, rain_split)            \
    "s_branch 9f\n\t" SMART_A_CASC_RAIN_OOL("9") "9:\n\t"

// ---- a chunk of four steps, threaded ---------------------------------------------------------------------------
// Three lanes of four arms each, calm / rain / dry.  A lane's step j ends with the dispatch of step j + 1, whose
// fall-through is the same lane; after step 3 the calm and the rain lane jump to the end of the chunk, the dry lane
// (laid out last: the most frequent kind) falls out of it.  The out-of-line cascades sit behind those two jumps.
// Labels: 10j / 11j / 12j = calm / dry / rain arm of step j; 130 = end of chunk.
// Operands beyond the arms': rn0..rn3, pe0..pe3 = the forcing of the four steps.
#define SMART_A_NEXT_FROM_CALM(j)                                                                                      \
    "s_cmp_eq_u64 %[rn" j "], 0\n\t"                                                                                   \
    "s_cbranch_scc0 12" j "f\n\t"                                                                                      \
    "s_cmp_eq_u64 %[pe" j "], 0\n\t"                                                                                   \
    "s_cbranch_scc0 11" j "f\n\t"
#define SMART_A_NEXT_FROM_RAIN(j)                                                                                      \
    "s_or_b64 %[tmp], %[rn" j "], %[pe" j "]\n\t"                                                                      \
    "s_cbranch_scc0 10" j "b\n\t"                                                                                      \
    "s_cmp_eq_u64 %[rn" j "], 0\n\t"                                                                                   \
    "s_cbranch_scc1 11" j "f\n\t"
#define SMART_A_NEXT_FROM_DRY(j)                                                                                       \
    "s_cmp_eq_u64 %[rn" j "], 0\n\t"                                                                                   \
    "s_cbranch_scc0 12" j "b\n\t"                                                                                      \
    "s_cmp_eq_u64 %[pe" j "], 0\n\t"                                                                                   \
    "s_cbranch_scc1 10" j "b\n\t"
#define SMART_A_CALM_J(route, j, deep, split) "10" j ":\n\t" SMART_A_CALM(route, SMART_A_CASC_CALM(j), deep, split)
#define SMART_A_DRY_J(route, j, split) "11" j ":\n\t" SMART_A_DRY(route, "%[pe" j "]", split)
#define SMART_A_RAIN_J(route, j, zeros, fill, deep, split)                                                             \
    "12" j ":\n\t" SMART_A_RAIN(route, "%[rn" j "]", "%[pe" j "]", SMART_A_CASC_RAIN(j), zeros, fill, deep, split)
// (alignment: the chunk -- its calm lane -- and the dry lane start on an 8-byte boundary, the rain lane at 4 mod 8; the two
// lie behind unconditional branches, their padding is never executed.  With the dispatches 16 bytes each, a dry step's
// nine instructions, a calm step's 45 behind its hook, and a rain step's first 10 and last 59 then lie on the boundary:
// the layout the fast builds of round 3 happened to have; a build whose chunk started at 4 mod 8 was 4.5 % slower)
#define SMART_A_CHUNK(route, deep, calm_split, zeros, drain, rain_split, dry_split)                                    \
    SMART_A_ALIGN8 SMART_A_NEXT_FROM_CALM("0") SMART_A_CALM_J(route, "0", deep, calm_split) SMART_A_NEXT_FROM_CALM("1")               \
    SMART_A_CALM_J(route, "1", deep, calm_split) SMART_A_NEXT_FROM_CALM("2")                                           \
    SMART_A_CALM_J(route, "2", deep, calm_split) SMART_A_NEXT_FROM_CALM("3")                                           \
    SMART_A_CALM_J(route, "3", deep, calm_split) "s_branch 130f\n\t"                                                   \
    SMART_A_CASC_CALM_OOL("0") SMART_A_CASC_CALM_OOL("1") SMART_A_CASC_CALM_OOL("2") SMART_A_CASC_CALM_OOL("3")        \
    SMART_A_ALIGN8_ODD                                                                                                 \
    SMART_A_RAIN_J(route, "0", zeros, SMART_A_FILL_QUICK(drain), deep, rain_split) SMART_A_NEXT_FROM_RAIN("1")         \
    SMART_A_RAIN_J(route, "1", zeros, SMART_A_FILL_QUICK(drain), deep, rain_split) SMART_A_NEXT_FROM_RAIN("2")         \
    SMART_A_RAIN_J(route, "2", zeros, SMART_A_FILL_QUICK(drain), deep, rain_split) SMART_A_NEXT_FROM_RAIN("3")         \
    SMART_A_RAIN_J(route, "3", zeros, SMART_A_FILL_QUICK(drain), deep, rain_split) "s_branch 130f\n\t"                 \
    SMART_A_CASC_RAIN_OOL("0") SMART_A_CASC_RAIN_OOL("1") SMART_A_CASC_RAIN_OOL("2") SMART_A_CASC_RAIN_OOL("3")        \
    SMART_A_ALIGN8                                                                                                     \
    SMART_A_DRY_J(route, "0", dry_split) SMART_A_NEXT_FROM_DRY("1") SMART_A_DRY_J(route, "1", dry_split)               \
    SMART_A_NEXT_FROM_DRY("2") SMART_A_DRY_J(route, "2", dry_split) SMART_A_NEXT_FROM_DRY("3")                         \
    SMART_A_DRY_J(route, "3", dry_split) "130:\n\t"
// not QUICK: the rain arm four times
#define SMART_A_CHUNK_RAIN(route, deep, zeros, drain, rain_split)                                                      \
    SMART_A_ALIGN8 SMART_A_RAIN_J(route, "0", zeros, SMART_A_FILL(drain), deep, rain_split)                                           \
    SMART_A_RAIN_J(route, "1", zeros, SMART_A_FILL(drain), deep, rain_split)                                           \
    SMART_A_RAIN_J(route, "2", zeros, SMART_A_FILL(drain), deep, rain_split)                                           \
    SMART_A_RAIN_J(route, "3", zeros, SMART_A_FILL(drain), deep, rain_split)                                           \
    "s_branch 130f\n\t" SMART_A_CASC_RAIN_OOL("0") SMART_A_CASC_RAIN_OOL("1") SMART_A_CASC_RAIN_OOL("2")               \
    SMART_A_CASC_RAIN_OOL("3") "130:\n\t"

// ---- a stretch of report intervals of the step loop as PAIRS of steps, each pair one straight-line block (round 4) -----
// What a lone wavefront pays for, measured (tools/microbench/lone.hip, profiles/r04_microbench_lone.txt): a vector or a
// scalar instruction 4 cycles, a conditional branch that is NOT taken 16, a taken one 24 to 40 -- the threaded chunk's
// dispatch (two compares, two branches not taken) costs a step as much as a dry step's nine vector instructions.
// Here the kind of every step is worked out ONCE per launch, by smart_forcing_scan: for each chunk of four steps two
// code words, one per pair of steps -- the byte offset of the block that holds the two arms of that pair, one behind the
// other with no dispatch in between.  Blocks lie SMART_P_STRIDE bytes apart (2 buffers x 2 pairs x 9 patterns + 2 x 2 for
// whole chunks of one kind; never-executed padding in between), a pair ends with  base + code word -> s_setpc_b64:  one
// computed jump per two steps instead of four branch instructions.  The forcing is loaded by the asm itself, two chunks
// ahead, into two fixed register buffers (the arms name their forcing by physical register; a buffer per chunk parity);
// the loop control (wait, request, count) sits in the tail of the second pair's blocks, and the report at the end of an
// interval in one block behind them all (SMART_P_REPORT): a whole stretch of intervals is ONE asm.  What a block knows
// about its first step lets the second drop work: a calm step behind a calm one has no demand pending (no compare, no
// hook), behind a dry one it has (the cascade in line, unconditionally).  The arithmetic is that of the threaded chunk
// and of Reporter::emit, operation for operation.
//   F0 = s[36:51], F1 = s[52:67]   forcing of the chunk at hand / the next one (rain, PE of step 0, 1, 2, 3)
//   s68, s69 / s70, s71            their code words (first pair, second pair)
//   s72 counter (pairs of chunks of the interval, counts up to zero), s75 the same for the intervals of the stretch,
//   s73 / s74 byte offsets of the last request into forcing / codes, s84 of the observation at hand (s[80:83]: it and
//   its deviation), s85: is the next report number 0?, s[76:77] jump target, s[78:79] address of block 0
#ifndef SMART_P_STRIDE
#error "SMART_P_STRIDE comes from smart_device.h"
#endif
#define SMART_P_ARM_C_N(route, id, deep, split) SMART_A_CALM(route, SMART_A_CASC_CALM(id), deep, split)
#define SMART_P_ARM_C_Q(route, deep, split) SMART_A_CALM_BODY(route, "", deep, split)
// (the cascade holds five 4-byte instructions: one s_nop puts what follows back on the 8-byte boundary)
#define SMART_P_ARM_C_F(route, deep, split) SMART_A_CALM_BODY(route, SMART_A_CASCADE "s_nop 0\n\t", deep, split)
#define SMART_P_ARM_R(route, rn, pe, id, zeros, drain, deep, split)                                                    \
    SMART_A_RAIN(route, rn, pe, SMART_A_CASC_RAIN(id), zeros, SMART_A_FILL_QUICK(drain), deep, split)
// ... behind a calm step nothing is pending (no test, no hook: 16 bytes less, the parity stays); behind a dry step every
// lane has a demand pending, and the wet ones -- there is one, past s_cbranch_execz -- take it now: the cascade in line
// (an s_nop ahead of it: the cascade's long runs on the 8-byte boundary, the filling's head behind it where it was)
#define SMART_P_ARM_R_Q(route, rn, pe, zeros, drain, deep, split)                                                      \
    SMART_A_RAIN_X(route, rn, pe, "", "", zeros, SMART_A_FILL_QUICK(drain), deep, split)
#define SMART_P_ARM_R_F(route, rn, pe, zeros, drain, deep, split)                                                      \
    SMART_A_RAIN_X(route, rn, pe, "", "s_nop 0\n\t" SMART_A_CASCADE, zeros, SMART_A_FILL_QUICK(drain), deep, split)
// tails: the jump to the second pair's block; the end of a chunk in buffer 0 / 1
// (no carry into the address's upper half: arm_intervals takes this path only when the code lies clear of a 4 GB line)
#define SMART_P_JUMP(code) "s_add_u32 s76, s78, " code "\n\ts_setpc_b64 s[76:77]\n\t"
#define SMART_P_REQUEST(f, c)                                                                                          \
    "s_add_u32 s73, s73, 64\n\t"                                                                                       \
    "s_load_dwordx16 " f ", %[fp], s73\n\t"                                                                            \
    "s_add_u32 s74, s74, 8\n\t"                                                                                        \
    "s_load_dwordx2 " c ", %[cp], s74\n\t"
#define SMART_P_TAIL_A0 SMART_P_JUMP("s69")
#define SMART_P_TAIL_A1 SMART_P_JUMP("s71")
#define SMART_P_TAIL_B0 "s_waitcnt lgkmcnt(0)\n\t" SMART_P_REQUEST("s[36:51]", "s[68:69]") SMART_P_JUMP("s70")
#define SMART_P_TAIL_B1                                                                                                \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                         \
    "s_add_u32 s72, s72, 1\n\t"                                                                                        \
    "s_cbranch_scc1 95f\n\t" SMART_P_REQUEST("s[52:67]", "s[70:71]") SMART_P_JUMP("s68")
// the nine blocks of one position: rx, px / ry, py = the forcing registers of its two steps.  A block whose first step is
// a rain step is entered 4 bytes behind its boundary (the code word says so): the rain arm wants to start at 4 mod 8
// (SMART_A_ALIGN8_ODD) -- and ends there (eleven 4-byte instructions on its way): behind a calm or a dry step it gets there
// by an s_nop, behind a rain step it is there, and a calm or a dry step behind a rain step gets an s_nop back
// block number n (position x 9 + pattern) starts SMART_P_STRIDE x n bytes behind block 0 (label 91): .org puts it there and
// refuses a predecessor that has outgrown its room.  The stride is NOT a power of two: with blocks 2 KB apart the k-th
// line of every block falls into one of 8 sets of the instruction cache
#define SMART_P_STR_(x) #x
#define SMART_P_STR(x) SMART_P_STR_(x)
#define SMART_P_BLOCK(pos, k, body) ".org 91b+(" pos "*9+" k ")*" SMART_P_STR(SMART_P_STRIDE) "\n\t" body
#define SMART_P_OOL(x) SMART_A_ALIGN8 x
#define SMART_P_NINE(pos, route, rx, px, ry, py, tail, deep, calm_split, zeros, drain, rain_split, dry_split)          \
    SMART_P_BLOCK(pos, "0", SMART_P_ARM_C_N(route, "0", deep, calm_split) SMART_P_ARM_C_Q(route, deep, calm_split)     \
                                tail SMART_P_OOL(SMART_A_CASC_CALM_OOL("0")))                                          \
    SMART_P_BLOCK(pos, "1", SMART_P_ARM_C_N(route, "0", deep, calm_split) SMART_A_DRY(route, py, dry_split)            \
                                tail SMART_P_OOL(SMART_A_CASC_CALM_OOL("0")))                                          \
    SMART_P_BLOCK(pos, "2", SMART_P_ARM_C_N(route, "0", deep, calm_split) "s_nop 0\n\t" SMART_P_ARM_R_Q(               \
        route, ry, py, zeros, drain, deep, rain_split) tail SMART_P_OOL(SMART_A_CASC_CALM_OOL("0")))                   \
    SMART_P_BLOCK(pos, "3", SMART_A_DRY(route, px, dry_split) SMART_P_ARM_C_F(route, deep, calm_split) tail)           \
    SMART_P_BLOCK(pos, "4", SMART_A_DRY(route, px, dry_split) SMART_A_DRY(route, py, dry_split) tail)                  \
    SMART_P_BLOCK(pos, "5", SMART_A_DRY(route, px, dry_split) "s_nop 0\n\t" SMART_P_ARM_R_F(                           \
        route, ry, py, zeros, drain, deep, rain_split) tail)                                                           \
    SMART_P_BLOCK(pos, "6", "s_nop 0\n\t" SMART_P_ARM_R(route, rx, px, "0", zeros, drain, deep, rain_split)            \
                                "s_nop 0\n\t" SMART_P_ARM_C_N(route, "1", deep, calm_split)                            \
                                    tail SMART_P_OOL(SMART_A_CASC_RAIN_OOL("0"))                                       \
                                        SMART_P_OOL(SMART_A_CASC_CALM_OOL("1")))                                       \
    SMART_P_BLOCK(pos, "7", "s_nop 0\n\t" SMART_P_ARM_R(route, rx, px, "0", zeros, drain, deep, rain_split)            \
                                "s_nop 0\n\t" SMART_A_DRY(route, py, dry_split)                                        \
                                    tail SMART_P_OOL(SMART_A_CASC_RAIN_OOL("0")))                                      \
    SMART_P_BLOCK(pos, "8", "s_nop 0\n\t" SMART_P_ARM_R(route, rx, px, "0", zeros, drain, deep, rain_split)            \
                                SMART_P_ARM_R(route, ry, py, "1", zeros, drain, deep, rain_split)                      \
                                    tail SMART_P_OOL(SMART_A_CASC_RAIN_OOL("0"))                                       \
                                        SMART_P_OOL(SMART_A_CASC_RAIN_OOL("1")))
// ... and two blocks per buffer for the chunks whose four steps are all calm or all dry (a night, a dry spell: 38 % of
// the chunks of the flat-forcing workload): the first code word leads there, the chunk's tail follows the fourth arm
#define SMART_P_QUADS(pos, kc, kd, route, p0, p1, p2, p3, tail, deep, calm_split, dry_split)                           \
    SMART_P_BLOCK(pos, kc, SMART_P_ARM_C_N(route, "0", deep, calm_split) SMART_P_ARM_C_Q(route, deep, calm_split)      \
                               SMART_P_ARM_C_Q(route, deep, calm_split) SMART_P_ARM_C_Q(route, deep, calm_split)       \
                                   tail SMART_P_OOL(SMART_A_CASC_CALM_OOL("0")))                                       \
    SMART_P_BLOCK(pos, kd, SMART_A_DRY(route, p0, dry_split) SMART_A_DRY(route, p1, dry_split)                         \
                               SMART_A_DRY(route, p2, dry_split) SMART_A_DRY(route, p3, dry_split) tail)
// The moments of a report about value `val` with observation e, deviation w (upper word whi: the missing mark): Reporter's
// (smart_device.h: emit / report_every), operation for operation
#define SMART_R_MOMENTS(val, e, w, whi)                                                                                \
    "s_cmp_eq_u32 " whi ", 0x7ff8dead\n\t"                                                                             \
    "s_cbranch_scc1 50f\n\t"                                                                                           \
    "s_nop 0\n\t"                                                                                                      \
    "v_add_f64 %[rd], " val ", -" e "\n\t"                                                                             \
    "v_add_f64 %[ru], " val ", -%[shift]\n\t"                                                                          \
    "v_add_f64 %[mA], %[mA], %[rd]\n\t"                                                                                \
    "v_fma_f64 %[mB], %[rd], %[rd], %[mB]\n\t"                                                                         \
    "v_add_f64 %[mC1], %[mC1], %[ru]\n\t"                                                                              \
    "v_fma_f64 %[mC2], %[ru], %[ru], %[mC2]\n\t"                                                                       \
    "v_fma_f64 %[mC3], " w ", %[ru], %[mC3]\n\t"                                                                       \
    "50:\n\t"
// The report at the end of an interval, in the asm (one copy per asm, reached from the tails of the last pair's blocks):
// nothing during the warm-up (%[rep] = 0); the value (value: the instruction that puts it into %[rv]); report 0 sets the
// constant the moments are taken about (s85: is the next report number 0?); the discharge through the lane's row pointer;
// the moments (e, w in s[80:83], requested when the interval began); `after`: what the run keeps of the interval.
#define SMART_P_REPORT(value, after) SMART_P_REPORT_X(value, after, "s[80:81]", "s[82:83]", "s83")
// (the discharge matrix is written once and read by another kernel, if at all: `nt` keeps it from displacing what the L2
// holds for the loop -- 1 % of a leg, profiles/r04_ab_store_and_icache.txt)
#ifndef SMART_STORE_NT
#define SMART_STORE_NT 1
#endif
#if SMART_STORE_NT
#define SMART_STORE_MOD " nt"
#else
#define SMART_STORE_MOD ""
#endif
#define SMART_P_REPORT_X(value, after, e, w, whi)                                                                      \
    "s_cmp_eq_u32 %[rep], 0\n\t"                                                                                       \
    "s_cbranch_scc1 96f\n\t" value "s_cmp_eq_u32 s85, 0\n\t"                                                           \
    "s_cbranch_scc1 93f\n\t"                                                                                           \
    "v_mov_b64_e64 %[shift], %[rv]\n\t"                                                                                \
    "s_mov_b32 s85, 0\n\t"                                                                                             \
    "93:\n\t"                                                                                                          \
    "s_cmp_eq_u32 %[sto], 0\n\t"                                                                                       \
    "s_cbranch_scc1 97f\n\t"                                                                                           \
    "global_store_dwordx2 %[row], %[rv], off" SMART_STORE_MOD "\n\t"                                                     \
    "v_lshl_add_u64 %[row], %[ld], 3, %[row]\n\t"                                                                      \
    "97:\n\t"                                                                                                          \
    "s_cmp_eq_u32 %[hob], 0\n\t"                                                                                       \
    "s_cbranch_scc1 98f\n\t" SMART_R_MOMENTS("%[rv]", e, w, whi) "98:\n\t" after "96:\n\t"
#define SMART_P_VALUE_MEAN "v_mul_f64 %[rv], %[acc], %[ig]\n\t"
#define SMART_P_AFTER_MEAN "v_add_f64 %[qtot], %[qtot], %[acc]\n\tv_mov_b64_e64 %[acc], 0\n\t"
#define SMART_P_VALUE_LAST "v_mov_b64_e64 %[rv], %[acc]\n\t"
#define SMART_P_AFTER_LAST "v_add_f64 %[numr], %[numr], %[qg]\n\tv_add_f64 %[denr], %[denr], %[qi]\n\t"
#define SMART_P_REPORT_MEAN SMART_P_REPORT(SMART_P_VALUE_MEAN, SMART_P_AFTER_MEAN)
#define SMART_P_REPORT_LAST SMART_P_REPORT(SMART_P_VALUE_LAST, SMART_P_AFTER_LAST)
// a STRETCH of %[niv] report intervals of %[half] pairs of chunks (EVEN) / chunks (ODD) each: %[fp] / %[cp] = the first chunk in the forcing /
// the code words, %[op] / %[wp] = the first interval's observation / deviation (anything readable without objective
// functions).  s75 counts the intervals, s84 is the byte offset of the observation at hand.
// (Two forms.  EVEN: intervals of an even number of chunks -- gaps that are multiples of eight steps: the interval ends in
// buffer 1, only its tails count, in pairs of chunks.  ODD: any whole number of chunks -- multiples of four: both
// buffers' tails count chunks, there is a report block behind each, and a stretch may start in either buffer: %[par].)
#define SMART_P_LOADS(fa, ca, fb, cb)                                                                                  \
    "s_load_dwordx16 " fa ", %[fp], 0x0\n\t"                                                                           \
    "s_load_dwordx2 " ca ", %[cp], 0x0\n\t"                                                                            \
    "s_mov_b32 s73, 64\n\t"                                                                                            \
    "s_load_dwordx16 " fb ", %[fp], s73\n\t"                                                                           \
    "s_mov_b32 s74, 8\n\t"                                                                                             \
    "s_load_dwordx2 " cb ", %[cp], s74\n\t"                                                                            \
    "s_mov_b32 s84, 0\n\t"                                                                                             \
    "s_load_dwordx2 s[80:81], %[op], 0x0\n\t"                                                                          \
    "s_load_dwordx2 s[82:83], %[wp], 0x0\n\t"                                                                          \
    "s_sub_u32 s72, 0, %[half]\n\t"                                                                                    \
    "s_sub_u32 s75, 0, %[niv]\n\t"                                                                                     \
    "s_mov_b32 s85, %[r0]\n\t"                                                                                         \
    "s_waitcnt lgkmcnt(0)\n\t"
#define SMART_P_ENTRY_EVEN SMART_P_LOADS("s[36:51]", "s[68:69]", "s[52:67]", "s[70:71]") SMART_P_JUMP("s68")
#define SMART_P_ENTRY_ODD                                                                                              \
    "s_cmp_eq_u32 %[par], 0\n\t"                                                                                       \
    "s_cbranch_scc0 88f\n\t" SMART_P_ENTRY_EVEN                                                                        \
    "88:\n\t" SMART_P_LOADS("s[52:67]", "s[70:71]", "s[36:51]", "s[68:69]") SMART_P_JUMP("s70")
#define SMART_P_TAIL_B0_ODD                                                                                            \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                         \
    "s_add_u32 s72, s72, 1\n\t"                                                                                        \
    "s_cbranch_scc1 94f\n\t" SMART_P_REQUEST("s[36:51]", "s[68:69]") SMART_P_JUMP("s70")
// what follows the report: the stretch over -> out; else the next interval's count, its observation, the request the
// tail did not make, on
#define SMART_P_NEXT_INTERVAL(f, c, code)                                                                              \
    "s_add_u32 s75, s75, 1\n\t"                                                                                        \
    "s_cbranch_scc1 99f\n\t"                                                                                           \
    "s_sub_u32 s72, 0, %[half]\n\t"                                                                                    \
    "s_add_u32 s84, s84, 8\n\t"                                                                                        \
    "s_load_dwordx2 s[80:81], %[op], s84\n\t"                                                                          \
    "s_load_dwordx2 s[82:83], %[wp], s84\n\t" SMART_P_REQUEST(f, c) SMART_P_JUMP(code)
#define SMART_P_REPORT94(report) "94:\n\t" report SMART_P_NEXT_INTERVAL("s[36:51]", "s[68:69]", "s70")
#define SMART_A_PAIRS_STRETCH_X(route, report, entry, tail_b0, block94, deep, calm_split, zeros, drain, rain_split,    \
                                dry_split)                                                                             \
    "s_getpc_b64 s[78:79]\n\t"                                                                                         \
    "90:\n\t"                                                                                                          \
    "s_add_u32 s78, s78, 91f-90b\n\t"                                                                                  \
    "s_addc_u32 s79, s79, 0\n\t"                                                                                       \
    "s_mov_b32 s77, s79\n\t" entry ".p2align 6\n\t"                                                                    \
    "91:\n\t" SMART_P_NINE("0", route, "s[36:37]", "s[38:39]", "s[40:41]", "s[42:43]", SMART_P_TAIL_A0, deep,          \
                           calm_split, zeros, drain, rain_split, dry_split)                                            \
        SMART_P_NINE("1", route, "s[44:45]", "s[46:47]", "s[48:49]", "s[50:51]", tail_b0, deep, calm_split, zeros,     \
                     drain, rain_split, dry_split)                                                                     \
            SMART_P_NINE("2", route, "s[52:53]", "s[54:55]", "s[56:57]", "s[58:59]", SMART_P_TAIL_A1, deep,            \
                         calm_split, zeros, drain, rain_split, dry_split)                                              \
                SMART_P_NINE("3", route, "s[60:61]", "s[62:63]", "s[64:65]", "s[66:67]", SMART_P_TAIL_B1, deep,        \
                             calm_split, zeros, drain, rain_split, dry_split)                                          \
                    SMART_P_QUADS("4", "0", "1", route, "s[38:39]", "s[42:43]", "s[46:47]", "s[50:51]", tail_b0, deep, \
                                  calm_split, dry_split)                                                               \
                        SMART_P_QUADS("4", "2", "3", route, "s[54:55]", "s[58:59]", "s[62:63]", "s[66:67]",            \
                                      SMART_P_TAIL_B1, deep, calm_split, dry_split) ".p2align 3\n\t" block94             \
    "95:\n\t" report SMART_P_NEXT_INTERVAL("s[52:67]", "s[70:71]", "s68") "99:\n\t"
#define SMART_A_PAIRS_STRETCH(route, report, deep, calm_split, zeros, drain, rain_split, dry_split)                    \
    SMART_A_PAIRS_STRETCH_X(route, report, SMART_P_ENTRY_EVEN, SMART_P_TAIL_B0, "", deep, calm_split, zeros, drain,    \
                            rain_split, dry_split)
#define SMART_A_PAIRS_STRETCH_ODD(route, report, deep, calm_split, zeros, drain, rain_split, dry_split)                \
    SMART_A_PAIRS_STRETCH_X(route, report, SMART_P_ENTRY_ODD, SMART_P_TAIL_B0_ODD, SMART_P_REPORT94(report), deep,     \
                            calm_split, zeros, drain, rain_split, dry_split)
#define SMART_P_CLOBBERS                                                                                               \
    "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67",    \
        "s70", "s71", "s72", "s73", "s74", "s76", "s77", "s78", "s79", "vcc", "scc"
#define SMART_S_CLOBBERS                                                                                               \
    "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51",    \
        "s68", "s69", "s75", "s80", "s81", "s82", "s83", "s84", "s85", SMART_P_CLOBBERS

// ---- a report every step (gap 1) as pair blocks with the report in the asm (round 4) ---------------------------------
// The every-step loop (time_loop_arms_each) dispatches step by step (SMART_A_STEP: one to two taken branches a step)
// and reports through compiled code between the asms.  Here smart_forcing_scan lays the run out as ONE stream of
// 32-byte records -- rain, PE, the observation of the step's report, its deviation from the mean -- and a code word per
// PAIR of steps; a pair is one s_load_dwordx16, two register buffers swap roles (P0 = s[36:51], P1 = s[52:67]; code
// words s68 / s70), and a block holds arm, report, arm, report and the loop control: one computed jump per two steps,
// no compiled code in between, for a whole stretch of steps (a multiple of four: the exit test sits in P1's tails).
// The report is Reporter's (smart_device.h: report_every), operation for operation: the discharge through a per-lane
// row pointer (lanes beyond the batch carry its last sample and store what its lane stores), the moments unless the
// deviation carries the missing-observation mark (its upper word: one scalar compare), the sum of the outflows.
// 18 blocks, SMART_E_STRIDE bytes apart.  After a rain arm -- which ends at 4 mod 8 -- and ahead of one an s_nop: the
// reports and the other arms on the 8-byte boundary.
#define SMART_E_STORE "global_store_dwordx2 %[row], %[acc], off" SMART_STORE_MOD "\n\tv_lshl_add_u64 %[row], %[ld], 3, %[row]\n\t"
#define SMART_E_MOMENTS(e, w, whi) SMART_R_MOMENTS("%[acc]", e, w, whi)
#define SMART_E_SUM "v_add_f64 %[qtot], %[qtot], %[acc]\n\t"
// rep(e, w, whi): the report text of a step (the caller composes it from the three pieces above)
#define SMART_E_C_N(rep, id) SMART_P_ARM_C_N(SMART_A_ROUTE_LAST, id, "", "") rep
#define SMART_E_C_Q(rep) SMART_P_ARM_C_Q(SMART_A_ROUTE_LAST, "", "") rep
#define SMART_E_C_F(rep) SMART_P_ARM_C_F(SMART_A_ROUTE_LAST, "", "") rep
#define SMART_E_D(pe, rep) SMART_A_DRY(SMART_A_ROUTE_LAST, pe, "") rep
#define SMART_E_R(rn, pe, id, rep) "s_nop 0\n\t" SMART_P_ARM_R(SMART_A_ROUTE_LAST, rn, pe, id, "", "", "", "") "s_nop 0\n\t" rep
#define SMART_E_R_Q(rn, pe, rep) "s_nop 0\n\t" SMART_P_ARM_R_Q(SMART_A_ROUTE_LAST, rn, pe, "", "", "", "") "s_nop 0\n\t" rep
#define SMART_E_R_F(rn, pe, rep) "s_nop 0\n\t" SMART_P_ARM_R_F(SMART_A_ROUTE_LAST, rn, pe, "", "", "", "") "s_nop 0\n\t" rep
#define SMART_E_BLOCK(pos, k, body) ".org 91b+(" pos "*9+" k ")*" SMART_P_STR(SMART_E_STRIDE) "\n\t" body
#define SMART_E_NINE(pos, rx, px, r0, ry, py, r1, tail)                                                                \
    SMART_E_BLOCK(pos, "0", SMART_E_C_N(r0, "0") SMART_E_C_Q(r1) tail SMART_P_OOL(SMART_A_CASC_CALM_OOL("0")))         \
    SMART_E_BLOCK(pos, "1", SMART_E_C_N(r0, "0") SMART_E_D(py, r1) tail SMART_P_OOL(SMART_A_CASC_CALM_OOL("0")))       \
    SMART_E_BLOCK(pos, "2", SMART_E_C_N(r0, "0") SMART_E_R_Q(ry, py, r1) tail SMART_P_OOL(SMART_A_CASC_CALM_OOL("0"))) \
    SMART_E_BLOCK(pos, "3", SMART_E_D(px, r0) SMART_E_C_F(r1) tail)                                                    \
    SMART_E_BLOCK(pos, "4", SMART_E_D(px, r0) SMART_E_D(py, r1) tail)                                                  \
    SMART_E_BLOCK(pos, "5", SMART_E_D(px, r0) SMART_E_R_F(ry, py, r1) tail)                                            \
    SMART_E_BLOCK(pos, "6", SMART_E_R(rx, px, "0", r0) SMART_E_C_N(r1, "1")                                            \
                                tail SMART_P_OOL(SMART_A_CASC_RAIN_OOL("0")) SMART_P_OOL(SMART_A_CASC_CALM_OOL("1")))  \
    SMART_E_BLOCK(pos, "7", SMART_E_R(rx, px, "0", r0) SMART_E_D(py, r1) tail SMART_P_OOL(SMART_A_CASC_RAIN_OOL("0"))) \
    SMART_E_BLOCK(pos, "8", SMART_E_R(rx, px, "0", r0) SMART_E_R(ry, py, "1", r1)                                      \
                                tail SMART_P_OOL(SMART_A_CASC_RAIN_OOL("0")) SMART_P_OOL(SMART_A_CASC_RAIN_OOL("1")))
#define SMART_E_REQUEST(p, c)                                                                                          \
    "s_add_u32 s73, s73, 64\n\t"                                                                                       \
    "s_load_dwordx16 " p ", %[sp], s73\n\t"                                                                            \
    "s_add_u32 s74, s74, 4\n\t"                                                                                        \
    "s_load_dword " c ", %[cp], s74\n\t"
#define SMART_E_TAIL_P0 "s_waitcnt lgkmcnt(0)\n\t" SMART_E_REQUEST("s[36:51]", "s68") SMART_P_JUMP("s70")
#define SMART_E_TAIL_P1                                                                                                \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                         \
    "s_add_u32 s72, s72, 1\n\t"                                                                                        \
    "s_cbranch_scc1 99f\n\t" SMART_E_REQUEST("s[52:67]", "s70") SMART_P_JUMP("s68")
// %[quads] = steps / 4 (>= 1); %[sp] / %[cp]: the first pair's record and code word; store / mom: SMART_E_STORE or "",
// SMART_E_MOMENTS or nothing
#define SMART_E_REP(store, mom, e, w, whi) store mom(e, w, whi) SMART_E_SUM
#define SMART_E_NOMOM(e, w, whi) ""
#define SMART_A_EVERY_STREAM(store, mom)                                                                               \
    "s_getpc_b64 s[78:79]\n\t"                                                                                         \
    "90:\n\t"                                                                                                          \
    "s_add_u32 s78, s78, 91f-90b\n\t"                                                                                  \
    "s_addc_u32 s79, s79, 0\n\t"                                                                                       \
    "s_mov_b32 s77, s79\n\t"                                                                                           \
    "s_load_dwordx16 s[36:51], %[sp], 0x0\n\t"                                                                         \
    "s_load_dword s68, %[cp], 0x0\n\t"                                                                                 \
    "s_mov_b32 s73, 64\n\t"                                                                                            \
    "s_load_dwordx16 s[52:67], %[sp], s73\n\t"                                                                         \
    "s_mov_b32 s74, 4\n\t"                                                                                             \
    "s_load_dword s70, %[cp], s74\n\t"                                                                                 \
    "s_sub_u32 s72, 0, %[quads]\n\t"                                                                                   \
    "s_waitcnt lgkmcnt(0)\n\t" SMART_P_JUMP("s68") ".p2align 6\n\t"                                                    \
    "91:\n\t" SMART_E_NINE("0", "s[36:37]", "s[38:39]", SMART_E_REP(store, mom, "s[40:41]", "s[42:43]", "s43"),        \
                           "s[44:45]", "s[46:47]", SMART_E_REP(store, mom, "s[48:49]", "s[50:51]", "s51"),             \
                           SMART_E_TAIL_P0)                                                                            \
        SMART_E_NINE("1", "s[52:53]", "s[54:55]", SMART_E_REP(store, mom, "s[56:57]", "s[58:59]", "s59"), "s[60:61]",  \
                     "s[62:63]", SMART_E_REP(store, mom, "s[64:65]", "s[66:67]", "s67"), SMART_E_TAIL_P1)              \
            ".p2align 3\n\t"                                                                                           \
            "99:\n\t"
#define SMART_E_CLOBBERS                                                                                               \
    "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51",    \
        "s68", SMART_P_CLOBBERS

// ---- report gaps that are not whole chunks (2, 3, 6 ... steps): the stream of records again (round 4) --------------------
// As SMART_A_EVERY_STREAM -- records of rain, PE, observation and deviation per step, a code word per pair of steps, two
// register buffers, the exit test in P1's tails -- but the report (SMART_P_REPORT_X: the interval's, with its flags) sits
// behind the arm whose step ends an interval, and nowhere else: the code word picks among three variants of every block
// (no report / behind the first arm / behind the second), smart_forcing_scan knows the gap.  2 x 3 x 9 = 54 blocks.
// An s_nop behind the report: eleven 4-byte instructions in it.
#define SMART_G_ARM1(arm, rep) arm rep
#define SMART_G_NINE(pos, v, route, rx, px, r0, ry, py, r1, tail)                                                      \
    SMART_E_BLOCK(pos, v "*9+0", SMART_P_ARM_C_N(route, "0", "", "") r0 SMART_P_ARM_C_Q(route, "", "") r1              \
                                     tail SMART_P_OOL(SMART_A_CASC_CALM_OOL("0")))                                     \
    SMART_E_BLOCK(pos, v "*9+1", SMART_P_ARM_C_N(route, "0", "", "") r0 SMART_A_DRY(route, py, "") r1                  \
                                     tail SMART_P_OOL(SMART_A_CASC_CALM_OOL("0")))                                     \
    SMART_E_BLOCK(pos, v "*9+2", SMART_P_ARM_C_N(route, "0", "", "") r0 "s_nop 0\n\t" SMART_P_ARM_R_Q(                 \
        route, ry, py, "", "", "", "") "s_nop 0\n\t" r1 tail SMART_P_OOL(SMART_A_CASC_CALM_OOL("0")))                  \
    SMART_E_BLOCK(pos, v "*9+3", SMART_A_DRY(route, px, "") r0 SMART_P_ARM_C_F(route, "", "") r1 tail)                 \
    SMART_E_BLOCK(pos, v "*9+4", SMART_A_DRY(route, px, "") r0 SMART_A_DRY(route, py, "") r1 tail)                     \
    SMART_E_BLOCK(pos, v "*9+5", SMART_A_DRY(route, px, "") r0 "s_nop 0\n\t" SMART_P_ARM_R_F(                          \
        route, ry, py, "", "", "", "") "s_nop 0\n\t" r1 tail)                                                          \
    SMART_E_BLOCK(pos, v "*9+6", "s_nop 0\n\t" SMART_P_ARM_R(route, rx, px, "0", "", "", "", "") "s_nop 0\n\t" r0      \
                                     SMART_P_ARM_C_N(route, "1", "", "") r1                                            \
                                         tail SMART_P_OOL(SMART_A_CASC_RAIN_OOL("0"))                                  \
                                             SMART_P_OOL(SMART_A_CASC_CALM_OOL("1")))                                  \
    SMART_E_BLOCK(pos, v "*9+7", "s_nop 0\n\t" SMART_P_ARM_R(route, rx, px, "0", "", "", "", "") "s_nop 0\n\t" r0      \
                                     SMART_A_DRY(route, py, "") r1 tail SMART_P_OOL(SMART_A_CASC_RAIN_OOL("0")))       \
    SMART_E_BLOCK(pos, v "*9+8", "s_nop 0\n\t" SMART_P_ARM_R(route, rx, px, "0", "", "", "", "") "s_nop 0\n\t" r0      \
                                     "s_nop 0\n\t" SMART_P_ARM_R(route, ry, py, "1", "", "", "", "") "s_nop 0\n\t" r1  \
                                         tail SMART_P_OOL(SMART_A_CASC_RAIN_OOL("0"))                                  \
                                             SMART_P_OOL(SMART_A_CASC_RAIN_OOL("1")))
#define SMART_G_REP(value, after, e, w, whi) SMART_P_REPORT_X(value, after, e, w, whi) "s_nop 0\n\t"
#define SMART_G_THREE(pos, route, value, after, rx, px, e0, w0, h0, ry, py, e1, w1, h1, tail)                          \
    SMART_G_NINE(pos, "0", route, rx, px, "", ry, py, "", tail)                                                        \
    SMART_G_NINE(pos, "1", route, rx, px, SMART_G_REP(value, after, e0, w0, h0), ry, py, "", tail)                     \
    SMART_G_NINE(pos, "2", route, rx, px, "", ry, py, SMART_G_REP(value, after, e1, w1, h1), tail)
// block number = position x 27 + variant x 9 + pattern: SMART_E_BLOCK's "pos*9 + k" with pos = 3 x position
#define SMART_A_GAP_STREAM(route, value, after)                                                                        \
    "s_getpc_b64 s[78:79]\n\t"                                                                                         \
    "90:\n\t"                                                                                                          \
    "s_add_u32 s78, s78, 91f-90b\n\t"                                                                                  \
    "s_addc_u32 s79, s79, 0\n\t"                                                                                       \
    "s_mov_b32 s77, s79\n\t"                                                                                           \
    "s_load_dwordx16 s[36:51], %[sp], 0x0\n\t"                                                                         \
    "s_load_dword s68, %[cp], 0x0\n\t"                                                                                 \
    "s_mov_b32 s73, 64\n\t"                                                                                            \
    "s_load_dwordx16 s[52:67], %[sp], s73\n\t"                                                                         \
    "s_mov_b32 s74, 4\n\t"                                                                                             \
    "s_load_dword s70, %[cp], s74\n\t"                                                                                 \
    "s_sub_u32 s72, 0, %[quads]\n\t"                                                                                   \
    "s_mov_b32 s85, %[r0]\n\t"                                                                                         \
    "s_waitcnt lgkmcnt(0)\n\t" SMART_P_JUMP("s68") ".p2align 6\n\t"                                                    \
    "91:\n\t" SMART_G_THREE("0", route, value, after, "s[36:37]", "s[38:39]", "s[40:41]", "s[42:43]", "s43",           \
                            "s[44:45]", "s[46:47]", "s[48:49]", "s[50:51]", "s51", SMART_E_TAIL_P0)                    \
        SMART_G_THREE("3", route, value, after, "s[52:53]", "s[54:55]", "s[56:57]", "s[58:59]", "s59", "s[60:61]",     \
                      "s[62:63]", "s[64:65]", "s[66:67]", "s67", SMART_E_TAIL_P1) ".p2align 3\n\t"                     \
                                                                                   "99:\n\t"

// ---- the wet interval of the interval engine (FastModel::wet_interval, merged regular variant, no exits) ----------
// `n` wet steps with one excess: 73 vector instructions a step (5 routing, 22 filling, 34 for the three leak passes
// with their powers, 8 for the layer sum / the balances, 4 reservoirs) and 2 scalar ones (counter, back-edge).
// hipcc's own loop over the same arithmetic carries 5 to 8 more per step (a 64-bit counter, a constant rebuilt in
// every turn, s_waitcnt's for loads that were long in) -- and a lone wavefront issues ONE instruction of any kind
// per turn of its SIMD.  s', s'^2, s'^3 in registers of their own here: `ex` and `eh` live across the steps.
// SMART_WET_E32 (round 6).  A wavefront that has its SIMD (nearly) to itself is fed by the instruction fetch: an 8-byte
// encoding costs it 5.6 cycles, a 4-byte one 4.7 (profiles/r05_microbench_lanes.txt, `probe`) -- and every fp64
// instruction is a VOP3 of 8 bytes except ONE: v_fmac_f64_e32  dst = src0 * src1 + dst, no modifiers.  22 of the wet
// step's 73 instructions are of that shape once the powers carry the sign (n_i = -s'^i: l = l * n_i + l is the leak
// fma(-l, s'^i, l), the same bits -- a product's magnitude does not depend on its factors' signs) and -D sits in a
// register of its own: the two products of the routing sum, the saturation excess' two shares, and the eighteen leaks.
// They come in runs of even length, so every 8-byte instruction starts where it did (SMART_A_WET_INTERVAL has the parities).
// The second leak pass forms its five factors first (four scratch registers more) and then leaks six times in a row.
#ifndef SMART_WET_E32
#define SMART_WET_E32 0
#endif
#if SMART_WET_E32
#define SMART_A_ROUTE_W                                                                                                \
    "v_mul_f64 %[t0], %[cg], %[yg]\n\t"                                                                                \
    "v_fmac_f64_e32 %[t0], %[cf], %[yf]\n\t"                                                                           \
    "v_fmac_f64_e32 %[t0], %[cs], %[ys]\n\t"                                                                           \
    "v_add_f64 %[acc], %[acc], %[riv]\n\t"                                                                             \
    "v_fma_f64 %[riv], %[riv], %[oma], %[t0]\n\t"
#define SMART_A_SAT_SHARES                                                                                             \
    "v_fmac_f64_e32 %[xf], %[npd], %[t1]\n\t"                                                                          \
    "v_fmac_f64_e32 %[xs], %[pd], %[t1]\n\t"
#define SMART_A_LEAK_N(l, p) "v_fmac_f64_e32 %[" l "], %[" l "], %[" p "]\n\t"
// s1 .. p6 hold -s', -s'^2, ... -s'^6 here
#define SMART_A_LEAKS_W                                                                                                \
    "v_mul_f64 %[s1], -%[sz], %[tot]\n\t"                                                                              \
    "v_mul_f64 %[p2], %[s1], -%[s1]\n\t"                                                                               \
    "v_mul_f64 %[p3], %[p2], -%[s1]\n\t"                                                                               \
    "v_mul_f64 %[p4], %[p2], -%[p2]\n\t"                                                                               \
    "v_mul_f64 %[p5], %[p4], -%[s1]\n\t"                                                                               \
    "v_mul_f64 %[p6], %[p3], -%[p3]\n\t" SMART_A_LEAK_N("l0", "s1") SMART_A_LEAK_N("l1", "p2")                         \
        SMART_A_LEAK_N("l2", "p3") SMART_A_LEAK_N("l3", "p4") SMART_A_LEAK_N("l4", "p5") SMART_A_LEAK_N("l5", "p6")    \
            SMART_A_LSUM("ai") "v_mul_f64 %[t1], %[s1], 0.5\n\t"                                                       \
                               "v_mul_f64 %[t0], %[s1], %[k3]\n\t"                                                     \
                               "v_ldexp_f64 %[xg], %[s1], -2\n\t"                                                      \
                               "v_mul_f64 %[q1], %[s1], %[k5]\n\t"                                                     \
                               "v_mul_f64 %[q2], %[s1], %[k6]\n\t" SMART_A_LEAK_N("l0", "s1") SMART_A_LEAK_N("l1", "t1") \
                SMART_A_LEAK_N("l2", "t0") SMART_A_LEAK_N("l3", "xg") SMART_A_LEAK_N("l4", "q1")                       \
                    SMART_A_LEAK_N("l5", "q2") SMART_A_LEAK_N("l0", "p6") SMART_A_LEAK_N("l1", "p5")                   \
                        SMART_A_LEAK_N("l2", "p4") SMART_A_LEAK_N("l3", "p3") SMART_A_LEAK_N("l4", "p2")               \
                            SMART_A_LEAK_N("l5", "s1")
#else
#define SMART_A_ROUTE_W SMART_A_ROUTE
#define SMART_A_SAT_SHARES                                                                                             \
    "v_fma_f64 %[xf], -%[pd], %[t1], %[xf]\n\t"                                                                        \
    "v_fma_f64 %[xs], %[pd], %[t1], %[xs]\n\t"
#define SMART_A_LEAKS_W SMART_A_LEAKS_("s1", "p2", "p3", "")
#endif
#define SMART_A_WET_HEAD SMART_A_ROUTE_W "v_fma_f64 %[xf], -%[eh], %[tot], %[ex]\n\t" SMART_A_FILL1("l0", "xf")
#define SMART_A_WET_FILL_TAIL                                                                                          \
    SMART_A_FILL1("l1", "t1") SMART_A_FILL1("l2", "t1") SMART_A_FILL1("l3", "t1") SMART_A_FILL1("l4", "t1")            \
        SMART_A_FILL1("l5", "t1") "v_mul_f64 %[xs], %[eh], %[tot]\n\t" SMART_A_SAT_SHARES
#define SMART_A_WET_REST                                                                                               \
    SMART_A_LEAKS_W "v_add_f64 %[t1], %[tot], -%[ai]\n\t"                                                              \
                    "v_add_f64 %[xf], %[xf], %[t1]\n\t" SMART_A_TOT_XG                                                 \
                    "v_fma_f64 %[ys], %[ys], %[ds], %[xs]\n\t"                                                         \
                    "v_fma_f64 %[yf], %[yf], %[df], %[xf]\n\t"                                                         \
                    "v_fma_f64 %[yg], %[yg], %[dg], %[xg]\n\t"                                                         \
                    "v_add_f64 %[xgs], %[xgs], %[xg]\n\t"
#define SMART_A_WET_STEP SMART_A_WET_HEAD SMART_A_WET_FILL_TAIL SMART_A_WET_REST
// SMART_WET_MODES 0: every step fills all six layers (round 3's first form, kept for the A/B).
// 1 / 2: two loops.  While the rain excess of every lane fits into the TOP layer -- 39 % of the wet wave-steps of the
// headline workload, 58 % of those of 6-hourly forcing, and within an interval a prefix of its steps: the top layer
// fills up, it does not empty under rain -- the other five layers see  t = l + 0; l = min(t, z); t - l = 0,  the
// identity as long as no layer is above its capacity (`ok`, wave-uniform, known at the start of the launch: nothing
// but a caller's initial state puts a layer there), and the saturation excess is zero: 15 instructions for a compare
// and a branch that is NOT taken.  The first step that leaves something over in any lane (or meets a NaN: v_cmp_nle)
// jumps into the middle of the loop of full steps and the interval ends there: one taken branch per interval, where
// an exit inside every step (the kernels with exits, FastModel::kExits) costs a lone wavefront one per step.
// Bit-identical to mode 0 (tools/debug/steps_bits.py).  2: both loops unrolled twice.
#ifndef SMART_WET_MODES
#define SMART_WET_MODES 3
#endif
#define SMART_A_WET_ABSORBED(fix)                                                                                      \
    SMART_A_WET_HEAD "v_cmp_nle_f64 vcc, %[t1], 0\n\t"                                                                 \
                     "s_cbranch_vccnz " fix "f\n\t"                                                                    \
                     "v_mul_f64 %[xs], %[eh], %[tot]\n\t" SMART_A_WET_REST
#define SMART_A_WET_FULL(fix) SMART_A_WET_HEAD fix ":\n\t" SMART_A_WET_FILL_TAIL SMART_A_WET_REST
#if SMART_WET_MODES == 0
#define SMART_A_WET_INTERVAL                                                                                           \
    "s_add_i32 %[cnt], %[n], 1\n\t"                                                                                    \
    "s_lshr_b32 %[cnt], %[cnt], 1\n\t"                                                                                 \
    "s_bitcmp1_b32 %[n], 0\n\t"                                                                                        \
    "s_cbranch_scc1 6f\n\t"                                                                                            \
    "5:\n\t" SMART_A_WET_STEP "6:\n\t" SMART_A_WET_STEP "s_add_i32 %[cnt], %[cnt], -1\n\t"                             \
    "s_cmp_lg_u32 %[cnt], 0\n\t"                                                                                       \
    "s_cbranch_scc1 5b\n\t"
#elif SMART_WET_MODES == 1
// cnt counts up from -n: s_add_u32 carries out (SCC) when it reaches zero.
// WHERE the 8-byte instructions lie counts (round 4).  Every fp64 instruction is a 64-bit encoding; one that starts at
// 4 mod 8 straddles a 32-byte fetch line every fourth time, and a wavefront that has its SIMD to itself waits for the
// second half (a same-instructions, same-registers shift of the step loop by 4 bytes moved its launch by 4.5 %:
// profiles/r04_placement_phases.txt).  Scalar instructions are 4 bytes, so the parity of a run of vector instructions is
// the parity of the number of scalar ones in front of it -- here, by hand:
//   the asm starts on an 8-byte boundary (SMART_A_ALIGN8: at most one s_nop, once per interval);
//   the three entry instructions put label 5 at 4 mod 8: the 10 instructions of the absorbed step's head straddle, the
//   s_cbranch_vccnz behind them puts its other 47 on the boundary, counter and back-edge (8 bytes) keep it there;
//   an s_nop behind `s_branch 9f` (never executed) puts label 6 -- the loop of full steps, 73 instructions and 8 bytes
//   of scalar ones a turn -- on the boundary as a whole.  Round 3's build had it at 4 mod 8 in the run loop.
#define SMART_A_WET_INTERVAL                                                                                           \
    SMART_A_ALIGN8 "s_sub_u32 %[cnt], 0, %[n]\n\t"                                                                     \
    "s_cmp_eq_u32 %[ok], 0\n\t"                                                                                        \
    "s_cbranch_scc1 6f\n\t"                                                                                            \
    "5:\n\t" SMART_A_WET_ABSORBED("7") "s_add_u32 %[cnt], %[cnt], 1\n\t"                                               \
    "s_cbranch_scc0 5b\n\t"                                                                                            \
    "s_branch 9f\n\t"                                                                                                  \
    "s_nop 0\n\t"                                                                                                      \
    "6:\n\t" SMART_A_WET_FULL("7") "s_add_u32 %[cnt], %[cnt], 1\n\t"                                                   \
    "s_cbranch_scc0 6b\n\t"                                                                                            \
    "9:\n\t"
#elif SMART_WET_MODES == 3
// FOUR steps per turn of either loop (round 4).  What a lone wavefront pays at the end of every step of mode 1 is a taken
// branch: 24 cycles and more (tools/microbench/lone.hip) on a step of 57 to 73 vector instructions; a branch that is not
// taken costs it 16, so unrolling with a test between the copies (mode 2) gains next to nothing.  Here the n mod 4
// steps that do not fill a turn come first, in mode 1's loops; the rest runs four to a turn with one counter and one
// back-edge per turn.  The absorbed loops leave for the full ones as in mode 1: from copy k into the middle of copy k.
// Same instructions per step, same bits.  Parities as in mode 1: an absorbed step starts at 4 mod 8 and ends on the
// boundary -- an s_nop between the copies of a turn (4 cycles against eleven straddling instructions); the full steps
// are 64-bit encodings throughout and start on the boundary.
#ifndef SMART_WET_TURN
#define SMART_WET_TURN 4 // steps per turn: 4 or 8
#endif
#if SMART_WET_TURN == 8
#define SMART_A_WET_SHIFT "3"
#define SMART_A_WET_MASK "7"
#define SMART_A_WET_ABS_MORE                                                                                           \
    "s_nop 0\n\t" SMART_A_WET_ABSORBED("75") "s_nop 0\n\t" SMART_A_WET_ABSORBED("76") "s_nop 0\n\t" SMART_A_WET_ABSORBED("77") \
    "s_nop 0\n\t" SMART_A_WET_ABSORBED("78")
#define SMART_A_WET_FULL_MORE SMART_A_WET_FULL("75") SMART_A_WET_FULL("76") SMART_A_WET_FULL("77") SMART_A_WET_FULL("78")
#else
#define SMART_A_WET_SHIFT "2"
#define SMART_A_WET_MASK "3"
#define SMART_A_WET_ABS_MORE ""
#define SMART_A_WET_FULL_MORE ""
#endif
#define SMART_A_WET_GROUPS                                                                                             \
    "s_lshr_b32 %[cnt], %[n], " SMART_A_WET_SHIFT "\n\t"                                                               \
    "s_sub_u32 %[cnt], 0, %[cnt]\n\t"                                                                                  \
    "s_cbranch_scc0 9f\n\t"
#define SMART_A_WET_INTERVAL                                                                                           \
    SMART_A_ALIGN8 "s_cmp_eq_u32 %[ok], 0\n\t"                                                                         \
    "s_cbranch_scc1 40f\n\t"                                                                                           \
    "s_and_b32 %[cnt], %[n], " SMART_A_WET_MASK "\n\t"                                                                 \
    "s_cbranch_scc0 20f\n\t"                                                                                           \
    "s_sub_u32 %[cnt], 0, %[cnt]\n\t"                                                                                  \
    "5:\n\t" SMART_A_WET_ABSORBED("7") "s_add_u32 %[cnt], %[cnt], 1\n\t"                                               \
    "s_cbranch_scc0 5b\n\t"                                                                                            \
    "20:\n\t" SMART_A_WET_GROUPS "25:\n\t" SMART_A_WET_ABSORBED("71") "s_nop 0\n\t" SMART_A_WET_ABSORBED("72")         \
    "s_nop 0\n\t" SMART_A_WET_ABSORBED("73") "s_nop 0\n\t" SMART_A_WET_ABSORBED("74") SMART_A_WET_ABS_MORE                   \
    "s_add_u32 %[cnt], %[cnt], 1\n\t"                                                                                  \
    "s_cbranch_scc0 25b\n\t"                                                                                           \
    "s_branch 9f\n\t"                                                                                                  \
    "40:\n\t"                                                                                                          \
    "s_and_b32 %[cnt], %[n], " SMART_A_WET_MASK "\n\t"                                                                 \
    "s_cbranch_scc0 30f\n\t"                                                                                           \
    "s_sub_u32 %[cnt], 0, %[cnt]\n\t"                                                                                  \
    "6:\n\t" SMART_A_WET_FULL("7") "s_add_u32 %[cnt], %[cnt], 1\n\t"                                                   \
    "s_cbranch_scc0 6b\n\t"                                                                                            \
    "30:\n\t" SMART_A_WET_GROUPS "s_nop 0\n\t"                                                                         \
    "31:\n\t" SMART_A_WET_FULL("71") SMART_A_WET_FULL("72") SMART_A_WET_FULL("73") SMART_A_WET_FULL("74")              \
    SMART_A_WET_FULL_MORE "s_add_u32 %[cnt], %[cnt], 1\n\t"                                                                                  \
    "s_cbranch_scc0 31b\n\t"                                                                                           \
    "9:\n\t"
#else
#define SMART_A_WET_INTERVAL                                                                                           \
    "s_sub_u32 %[cnt], 0, %[n]\n\t"                                                                                    \
    "s_cmp_eq_u32 %[ok], 0\n\t"                                                                                        \
    "s_cbranch_scc1 6f\n\t"                                                                                            \
    "5:\n\t" SMART_A_WET_ABSORBED("7") "s_add_u32 %[cnt], %[cnt], 1\n\t"                                               \
    "s_cbranch_scc1 9f\n\t" SMART_A_WET_ABSORBED("8") "s_add_u32 %[cnt], %[cnt], 1\n\t"                                \
    "s_cbranch_scc0 5b\n\t"                                                                                            \
    "s_branch 9f\n\t"                                                                                                  \
    "6:\n\t" SMART_A_WET_FULL("7") "s_add_u32 %[cnt], %[cnt], 1\n\t"                                                   \
    "s_cbranch_scc1 9f\n\t" SMART_A_WET_FULL("8") "s_add_u32 %[cnt], %[cnt], 1\n\t"                                    \
    "s_cbranch_scc0 6b\n\t"                                                                                            \
    "9:\n\t"
#endif
// ... and the calm interval: `n` wet steps with ZERO excess (no rain, no evaporation: the night block of sub-daily
// data) need no filling -- with no layer above capacity it is the identity, as in the calm arm of the step loop: 51
// instructions a step instead of 73.
#define SMART_A_CALM_STEP                                                                                              \
    SMART_A_ROUTE SMART_A_LEAKS_("s1", "p2", "p3", "") "v_add_f64 %[xf], %[tot], -%[ai]\n\t" SMART_A_TOT_XG            \
    "v_fma_f64 %[ys], %[ys], %[ds], 0\n\t"                                                                             \
    "v_fma_f64 %[yf], %[yf], %[df], %[xf]\n\t"                                                                         \
    "v_fma_f64 %[yg], %[yg], %[dg], %[xg]\n\t"                                                                         \
    "v_add_f64 %[xgs], %[xgs], %[xg]\n\t"
#define SMART_A_CALM_INTERVAL                                                                                          \
    SMART_A_ALIGN8 "s_add_i32 %[cnt], %[n], 1\n\t"                                                                                    \
    "s_lshr_b32 %[cnt], %[cnt], 1\n\t"                                                                                 \
    "s_bitcmp1_b32 %[n], 0\n\t"                                                                                        \
    "s_cbranch_scc1 6f\n\t"                                                                                            \
    "5:\n\t" SMART_A_CALM_STEP "6:\n\t" SMART_A_CALM_STEP "s_add_i32 %[cnt], %[cnt], -1\n\t"                           \
    "s_cmp_lg_u32 %[cnt], 0\n\t"                                                                                       \
    "s_cbranch_scc1 5b\n\t"
