"""The ALGORITHM of the row form of the literal step (smartpy_amd/csrc/smart_literal_lanes.h, round 5), restated lane by
lane in Python with exactly rounded arithmetic, against the oracle's one-step on the CPU -- no GPU needed.

The HIP code spreads one sample over the sixteen lanes of a DPP row: six soil layers in lanes 0..5, five reservoirs in
lanes 0 / 4 / 8 / 12 / 13, sums as chains of fma(x[lane k], 1.0, acc) in the reference's order, the two cascades as
chains with per-lane masks (e = max(fma(sp[k], m, e), 0)), divisions by per-sample constants as reciprocal +
correction step.  Whether THAT arrangement of operations gives the reference's bits is a question about arithmetic,
not about the GPU, and can be checked here: every operation below is the one the asm / HIP code performs, on sixteen
Python floats per register, with fma() computed exactly (rational arithmetic, one rounding).  The GPU suite then holds
the compiled code to the same oracle (tests/test_gpu_parity.py); tests/test_lanes_isa.py looks at its instruction
stream.
"""
from fractions import Fraction

import numpy as np
import pytest

from oracle import smart_oracle as so, lhs_oracle

L = 16
OVE, DRA, INT, SGW, DGW = 12, 13, 8, 0, 4        # lanes of the reservoirs (LiteralLanesModel::kOve ...)


def fma(a, b, c):
    """RN(a * b + c), one rounding (float(Fraction) is correctly rounded)."""
    r = Fraction(a) * Fraction(b) + Fraction(c)
    if r == 0:      # the sign of an exact zero sum: +0 unless both addends are -0 (round to nearest)
        prod_neg = (np.signbit(a) != np.signbit(b))
        return -0.0 if (prod_neg and np.signbit(c)) else 0.0
    return float(r)


def vmax(a, b):
    """v_max_f64: the larger; of two zeros the positive one."""
    if a == b:
        return a if not np.signbit(a) else b
    return a if a > b else b


def vmin(a, b):
    """v_min_f64: the smaller; of two zeros the negative one."""
    if a == b:
        return a if np.signbit(a) else b
    return a if a < b else b


def dv(a, b, y):
    """Slow::dv<true>: a / b through y = RN(1 / b) and one correction step."""
    q0 = a * y
    r = fma(-b, q0, a)
    return fma(r, y, q0)


class Row(object):
    """One sample in sixteen lanes: every register is a list of sixteen floats, every operation acts on all of them."""

    def __init__(self, area, dt, p, states):
        self.area, self.dt = area, dt
        self.pT, self.pC, self.pH, self.pD, self.pS, self.pZ = p[:6]
        sk, fk, gk = p[6] * 3600.0, p[7] * 3600.0, p[8] * 3600.0
        self.rk = p[9] * 3600.0
        self.y_area, self.y_z, self.y_dt, self.y_rk = 1.0 / area, 1.0 / self.pZ, 1.0 / dt, 1.0 / self.rk
        self.z = self.pZ / 6.0
        omd = 1.0 - self.pD
        r = range(L)
        self.kres = [sk if i in (OVE, DRA) else fk if i == INT else gk if i in (SGW, DGW) else 2.0 ** 500 for i in r]
        # m[k][lane], cm[k][lane]: the chain is "above" lane i's layer while k < i; lanes that are no layer see all of it,
        # the overland lane none of it
        above = [[(i != OVE) if i >= 6 else (k < i) for i in r] for k in range(6)]
        self.m = [[-1.0 if a else 0.0 for a in row] for row in above]
        self.cm = [[self.pC if a else 1.0 for a in row] for row in above]
        self.w_of = [0.0 if i == OVE else 1.0 for i in r]
        self.sx = [1.0 if i == OVE else 0.0 for i in r]
        self.kx = [self.pD if i == DRA else omd if i == INT else 0.0 for i in r]
        mul = [[(k <= i) if i < 6 else (k <= 5 - (i - 8)) if 8 <= i < 14 else False for i in r] for k in range(1, 6)]
        self.dk = [[1.0 if x else 0.0 for x in row] for row in mul]
        self.ek = [[0.0 if x else 1.0 for x in row] for row in mul]
        self.nby = [float(i + 1) if i < 6 else 1.0 for i in r]
        self.by = [1.0 / n for n in self.nby]
        v = dict(zip((OVE, DRA, INT, SGW, DGW), states[:5]))
        self.V = [v.get(i, 0.0) for i in r]
        self.LY = [states[5 + i] if i < 6 else 0.0 for i in r]
        self.VRIV = states[11]

    def step(self, rain_in, peva_in):
        r = range(L)
        area, dt = self.area, self.dt
        l = [dv(self.LY[i], area, self.y_area) * 1e3 for i in r]
        tot = l[0]                                             # v_mov_b64_dpp row_newbcast:0, then five fmac chains
        for k in range(1, 6):
            tot = fma(l[k], 1.0, tot)
        rain = rain_in * self.pT
        ex0 = rain - peva_in
        if ex0 >= 0.0:
            aeva = peva_in
            q = dv(tot, self.pZ, self.y_z)
            hp = self.pH * q
            of = hp * ex0
            e = [fma(-of, self.w_of[i], ex0) for i in r]
            s1 = self.pS * q
            sp = [self.z - l[i] for i in r]
            for k in range(6):                                 # the filling chain: sp of lane k to every lane
                e = [vmax(fma(sp[k], self.m[k][i], e[i]), 0.0) for i in r]
            l = [l[i] + e[i] if e[i] <= sp[i] else self.z for i in r]
            X = [fma(hp, self.sx[i], self.kx[i]) * e[i] for i in r]
            P = [s1] * L
            for k in range(5):
                P = [P[i] * fma(s1, self.dk[k][i], self.ek[k][i]) for i in r]
            f2 = [dv(s1, self.nby[i], self.by[i]) for i in r]
            PC = [P[i + 8] if i < 8 else P[i] for i in r]      # v_mov_b32_dpp row_shl:8 (lanes 8..15 keep theirs)

            def leak(f, lanes, order):                          # x = l f; l -= x; the total into the lanes of ONE bank
                nonlocal l, X
                x = [l[i] * f[i] for i in r]
                l = [l[i] - x[i] for i in r]
                for k in order:
                    X = [fma(x[k], 1.0, X[i]) if i in lanes else X[i] for i in r]
            leak(P, range(8, 12), range(6))                     # interflow: bank 2
            leak(f2, range(0, 4), range(6))                     # shallow groundwater: bank 0
            leak(PC, range(4, 8), range(5, -1, -1))             # deep groundwater, bottom layer first: bank 1
        else:
            aeva = rain
            d = [ex0 * -1.0] * L
            for k in range(6):                                  # the evaporation chain
                d = [self.cm[k][i] * vmax(fma(l[k], self.m[k][i], d[i]), 0.0) for i in r]
            take = [vmin(l[i], d[i]) for i in r]
            l = [l[i] - take[i] for i in r]
            X = [0.0] * L
            for k in range(6):
                aeva = fma(take[k], 1.0, aeva)
        out0 = dv(dv(aeva, 1e3, 1e-3) * area, dt, self.y_dt)
        outq = [self.V[i] / self.kres[i] for i in r]            # a true division
        self.V = [vmax(self.V[i] + (dv(X[i], 1e3, 1e-3) * area - outq[i] * dt), 0.0) for i in r]
        self.LY = [dv(l[i], 1e3, 1e-3) * area for i in r]
        q_in = outq[OVE]
        for k in (DRA, INT, SGW, DGW):
            q_in = fma(outq[k], 1.0, q_in)
        v_old = self.VRIV                                        # river_q<true>
        q = dv(v_old, self.rk, self.y_rk)
        tmp = v_old + (q_in - q) * dt
        if tmp < 0.0:
            q = 0.95 * (q_in + dv(v_old, dt, self.y_dt))
            self.VRIV = v_old + (q_in - q) * dt
        else:
            self.VRIV = tmp
        return np.array([out0, outq[OVE], outq[DRA], outq[INT], outq[SGW], outq[DGW], q,
                         self.V[OVE], self.V[DRA], self.V[INT], self.V[SGW], self.V[DGW]] + self.LY[:6] + [self.VRIV])


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.int64)


@pytest.mark.parametrize('dt', [86400.0, 3600.0])
def test_the_row_forms_arithmetic_is_the_references(dt):
    """40 parameter rows x 60 steps each (wet and dry, full and empty layers, the river's 95 % rule for the daily rows
    with dt / RK > 2), every one of the 19 outputs of every step: the oracle's bits.  The rows satisfy what the row
    form asks of a quick chunk (default ranges: C >= 0, H <= 0.99, S <= 0.74; states positive normal numbers)."""
    rng = np.random.default_rng(int(dt))
    params = lhs_oracle.lhs_params(40, seed=int(dt) % 977)
    if dt == 86400.0:
        params[::2, 9] = rng.uniform(1.0, 11.0, 20)              # ill-conditioned rows: the ones that run this form
    area = 175.46e6
    n_checked = wet = dry = fired = 0
    for p in params:
        st = np.concatenate([rng.uniform(1e3, 5e6, 5), rng.uniform(0.0, 1.0, 6) * (p[5] / 6.0) / 1000.0 * area,
                             [rng.uniform(1e3, 1e6)]])
        st[5 + rng.integers(0, 6)] = 0.0                         # an empty layer, a full one
        st[5 + rng.integers(0, 6)] = (p[5] / 6.0) / 1000.0 * area
        row = Row(area, dt, p, st)
        for t in range(60):
            rain = float(rng.gamma(0.7, 4.5) * (rng.random() < 0.6)) * dt / 86400.0
            peva = float(rng.uniform(0.0, 3.0)) * dt / 86400.0
            want = so.one_step(area, dt, rain, peva, p, st, pow_mode=so.POW_MUL)
            v_riv_before = st[11]
            got = row.step(rain, peva)
            assert np.array_equal(_bits(got), _bits(want)), (p, t, np.nonzero(_bits(got) != _bits(want))[0])
            st = want[7:19].copy()
            n_checked += 1
            wet += rain * p[0] - peva >= 0.0
            dry += rain * p[0] - peva < 0.0
            fired += want[6] != v_riv_before / (p[9] * 3600.0)
    assert n_checked == 2400 and wet > 400 and dry > 400
    assert (fired > 50) == (dt == 86400.0)                      # the 95 % rule fired where the river is ill-conditioned
