"""Time axes and temporal resampling on the host (counterpart of smartpy/timeframe.py).

This runs once per catchment, before any sample is simulated, and produces the arrays the GPU path
consumes: forcing per simulation step and observations per report step.  The reference builds dictionaries
keyed by datetime and walks them stamp by stamp; here every series lives on an integer grid of seconds
and is resampled with numpy.  Arithmetic that reaches the model -- the equal split of a cumulative value
over sub-steps and the re-aggregation (timeframe.py:167-233), the backward replication of mean values and
their re-averaging (timeframe.py:236-309) -- is done in the reference's operation order (each total starts
at 0.0 and adds the stamps from the latest backwards), so the arrays are bit-identical to the reference's
(tests/test_host_logic.py against tests/golden/forcing_example.npz).
"""
from datetime import datetime, timedelta
from math import gcd

import numpy as np

_EPOCH = datetime(1970, 1, 1)


def _sec(dt):
    """datetime -> integer seconds since the epoch (exact for whole-second stamps)."""
    d = dt - _EPOCH
    return d.days * 86400 + d.seconds


def _isec(delta):
    return int(delta.total_seconds())


class TimeFrame(object):
    """Simulation ('simu') and reporting ('save') time axes, each with one leading stamp for the initial
    conditions (timeframe.py:50-115).  len(simu_series) - 1 is the number of model steps T and
    len(save_series) - 1 the number of report steps R (structure.py:73-75)."""

    def __init__(self, dt_save_start, dt_save_end, simu_increment, save_increment):
        self.save_start = dt_save_start
        self.save_gap = save_increment
        self.save_end = self._check_save_end(dt_save_end)
        self.simu_gap = simu_increment
        self.simu_start, self.simu_end = self._get_simu_start_end_given_save_start_end()
        self.save_series = self._series(self.save_start - self.save_gap, self.save_end, self.save_gap)
        self.simu_series = self._series(self.simu_start - self.simu_gap, self.simu_end, self.simu_gap)

    @staticmethod
    def _series(first, last, gap):
        n = (_sec(last) - _sec(first)) // _isec(gap) + 1 if last >= first else 0
        return [first + k * gap for k in range(n)]

    def _check_save_end(self, save_end):
        if not self.save_start <= save_end:
            raise Exception("Save Start is greater than Save End.")
        divisor, remainder = divmod(int((save_end - self.save_start).total_seconds()), _isec(self.save_gap))
        if remainder != 0:
            latest = self.save_start + timedelta(seconds=self.save_gap.total_seconds()) * divisor
            raise Exception("The combination of (start, end) datetimes and the saving time delta "
                            "are not compatible. For a start at {}, and a time delta of {}, the "
                            "latest end in the period is {}".format(self.save_start, self.save_gap, latest))
        return save_end

    def _get_simu_start_end_given_save_start_end(self):
        if not self.save_gap.total_seconds() % self.simu_gap.total_seconds() == 0:
            raise Exception("Save Gap is not greater and a multiple of Simulation Gap.")
        return self.save_start - self.save_gap + self.simu_gap, self.save_end

    def get_gap_simu(self):
        return self.simu_gap

    def get_gap_report(self):
        return self.save_gap

    def get_series_simu(self):
        return self.simu_series

    def get_series_save(self):
        return self.save_series

    # sizes the engine needs
    @property
    def n_steps(self):
        return len(self.simu_series) - 1

    @property
    def n_reports(self):
        return len(self.save_series) - 1

    @property
    def report_gap(self):
        return self.n_steps // self.n_reports


def check_interval_in_list(list_of_dt, csv_file):
    """Regular spacing and no missing stamp (timeframe.py:144-155) -> (first, last, interval)."""
    secs = np.array([_sec(d) for d in list_of_dt], dtype=np.int64)
    steps = np.unique(np.diff(secs))
    if len(steps) == 1:
        if secs[0] + steps[0] * (len(secs) - 1) == secs[-1]:
            return list_of_dt[0], list_of_dt[-1], timedelta(seconds=int(steps[0]))
        raise Exception('Missing Data: {} is missing at least one datetime in period.'.format(csv_file))
    raise Exception('Inconsistent Interval: {} does not feature a single time interval.'.format(csv_file))


def get_required_resolution(start_data, start_simu, delta_data, delta_simu):
    """Finest grid on which both the data stamps and the simulation stamps fall (timeframe.py:158-164)."""
    return timedelta(seconds=gcd(int((start_data - start_simu).total_seconds()),
                                 gcd(_isec(delta_data), _isec(delta_simu))))


def resample_cumulative(values, start_data, delta_data, delta_res, start_simu, end_simu, delta_simu):
    """Cumulative series (rain, PE; value at stamp t = amount over (t - delta_data, t]) -> amounts per
    simulation step for the stamps start_simu .. end_simu (timeframe.py:167-233).

    Two stages like the reference: every data value is split equally over its delta_data / delta_res
    sub-steps, then each simulation stamp sums its delta_simu / delta_res sub-steps, latest first.
    """
    values = np.asarray(values, dtype=np.float64)
    res = _isec(delta_res)
    div_data, rem = divmod(_isec(delta_data), res)
    if rem != 0:
        raise Exception("Increase Resolution: Time Deltas are not multiples of each other.")
    div_simu, rem = divmod(_isec(delta_simu), res)
    if rem != 0:
        raise Exception("Decrease Resolution: Time Deltas are not multiples of each other.")
    if div_data < 1:
        raise Exception("Increase Resolution: Low resolution lower than higher resolution "
                        "{} < {}.".format(delta_data, delta_res))
    if div_simu < 1:
        raise Exception("Decrease Resolution: Low resolution lower than higher resolution "
                        "{} < {}.".format(delta_simu, delta_res))
    # fine grid: index i <-> time t0 + i * res, t0 = first fine stamp = start_data - (div_data - 1) * res
    fine = np.repeat(values / div_data, div_data) if div_data > 1 else values
    t0 = _sec(start_data) - (div_data - 1) * res
    n_simu = (_sec(end_simu) - _sec(start_simu)) // _isec(delta_simu) + 1
    last = (_sec(start_simu) - t0) // res + np.arange(n_simu, dtype=np.int64) * div_simu
    if (_sec(start_simu) - t0) % res != 0 or last[0] - (div_simu - 1) < 0 or last[-1] >= len(fine):
        raise KeyError('data do not cover the simulation period')
    out = np.zeros(n_simu, dtype=np.float64)
    for j in range(div_simu):           # 0.0 + fine[t] + fine[t - res] + ...  (timeframe.py:204-206)
        out += fine[last - j]
    return out


def resample_irregular_mean(stamps, values, start_report, end_report, delta_lo, delta_hi):
    """Mean-valued observations at possibly irregular stamps (daily mean flows with gaps) -> one mean per
    report stamp start_report .. end_report every delta_lo, NaN where any sub-step is missing
    (timeframe.py:236-309: backward replication onto a delta_hi grid, then arithmetic mean of the
    delta_lo / delta_hi sub-steps ending at each report stamp, latest first).
    """
    hi, lo = _isec(delta_hi), _isec(delta_lo)
    div_lo, rem = divmod(lo, hi)
    if rem != 0:
        raise Exception("Decrease Resolution: Time Deltas are not multiples of each other.")
    if div_lo < 1:
        raise Exception("Decrease Resolution: Low resolution lower than higher resolution "
                        "{} < {}.".format(delta_lo, delta_hi))
    n_rep = (_sec(end_report) - _sec(start_report)) // lo + 1
    out = np.full(n_rep, np.nan, dtype=np.float64)
    if len(stamps) == 0:
        return out
    secs = np.array([_sec(d) for d in stamps], dtype=np.int64)
    values = np.asarray(values, dtype=np.float64)
    # span of each observation: back to the previous stamp, or one standard interval when the previous one is
    # 1.5 intervals or more away (a gap) or does not exist (timeframe.py:246-254)
    span = np.diff(secs, prepend=secs[0] - lo)
    span = np.where(span >= 1.5 * lo, lo, span)
    if np.any(span % hi != 0):
        raise Exception("Increase Resolution: Time Deltas are not multiples of each other.")
    if np.any(span < hi):
        raise Exception("Increase Resolution: Low resolution lower than higher resolution "
                        "{} < {}.".format(timedelta(seconds=int(span.min())), delta_hi))
    # fine grid anchored on the report axis; observations whose stamps do not fall on it never meet a report
    # sub-step (the reference's dictionary lookup misses them, timeframe.py:294)
    g0 = _sec(start_report) - (div_lo - 1) * hi
    aligned = (secs - g0) % hi == 0
    n_fine = (n_rep - 1) * div_lo + div_lo
    fine = np.full(n_fine, np.nan, dtype=np.float64)
    have = np.zeros(n_fine, dtype=bool)
    reps = (span // hi).astype(np.int64)
    idx_last = (secs - g0) // hi
    for k in np.nonzero(aligned)[0]:
        i1 = idx_last[k]
        i0 = i1 - reps[k] + 1
        lo_i, hi_i = max(i0, 0), min(i1, n_fine - 1)
        if lo_i > hi_i:
            continue
        seg = slice(lo_i, hi_i + 1)
        if np.any(have[seg] & (fine[seg] != 0.0)):      # a truthy value is already there (timeframe.py:270-271)
            raise Exception("Increase Resolution: Overwriting already existing data for datetime.")
        fine[seg] = values[k]
        have[seg] = True
    last = (div_lo - 1) + np.arange(n_rep, dtype=np.int64) * div_lo
    acc = np.zeros(n_rep, dtype=np.float64)
    ok = np.ones(n_rep, dtype=bool)
    for j in range(div_lo):             # 0.0 + v[t] + v[t - hi] + ...  then / divisor (timeframe.py:290-293)
        acc += np.where(have[last - j], fine[last - j], 0.0)
        ok &= have[last - j]
    out[ok] = acc[ok] / div_lo
    return out
