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
import argparse
from collections import OrderedDict
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

    def _get_list_save_dt_with_initial_conditions(self):
        """timeframe.py:95-104: report stamps with one leading stamp for the initial conditions."""
        return self._series(self.save_start - self.save_gap, self.save_end, self.save_gap)

    def _get_list_simu_dt_with_initial_conditions(self):
        """timeframe.py:106-115."""
        return self._series(self.simu_start - self.simu_gap, self.simu_end, self.simu_gap)

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


# ----------------------------------------------------------------------------------------------------------
# The reference's dictionary-keyed helpers (timeframe.py:130-141, 167-309), for scripts that call them directly.
# Same names, arguments, results and error texts; the work is done by the array routines above, so the values are
# the ones the GPU path is fed with.
# ----------------------------------------------------------------------------------------------------------
def valid_date(s):
    """argparse type 'dd/mm/YYYY_HH:MM:SS' (timeframe.py:130-134)."""
    try:
        return datetime.strptime(s, "%d/%m/%Y_%H:%M:%S")
    except ValueError:
        raise argparse.ArgumentTypeError("Not a valid date: '{0}'.".format(s))


def valid_delta_min(n):
    """argparse type: minutes -> timedelta (timeframe.py:137-141)."""
    try:
        return timedelta(minutes=int(n))
    except ValueError:
        raise argparse.ArgumentTypeError("Not a valid time delta: '{0}'.".format(n))


def _divisor(delta_lo, delta_hi, what):
    divisor, remainder = divmod(_isec(delta_lo), _isec(delta_hi))
    if remainder != 0:
        raise Exception("{} Resolution: Time Deltas are not multiples of each other.".format(what))
    if divisor < 1:
        raise Exception("{} Resolution: Low resolution lower than higher resolution "
                        "{} < {}.".format(what, delta_lo, delta_hi))
    return divisor


def increase_time_resolution_of_regular_cumulative_data(dict_info, start_lo, end_lo, time_delta_lo, time_delta_hi):
    """Each value split equally over its sub-steps, written backwards from its stamp (timeframe.py:167-186)."""
    divisor = _divisor(time_delta_lo, time_delta_hi, "Increase")
    stamps = TimeFrame._series(start_lo, end_lo, time_delta_lo)
    portions = np.array([dict_info[t] for t in stamps], dtype=np.float64) / divisor
    out = dict()
    for t, v in zip(stamps, portions):
        for k in range(divisor):
            out[t - k * time_delta_hi] = v
    return out


def decrease_time_resolution_of_regular_cumulative_data(dict_info, start_lo, end_lo, time_delta_lo, time_delta_hi):
    """Each low-resolution stamp sums its sub-steps, latest first (timeframe.py:189-208)."""
    divisor = _divisor(time_delta_lo, time_delta_hi, "Decrease")
    stamps = TimeFrame._series(start_lo, end_lo, time_delta_lo)
    total = np.zeros(len(stamps), dtype=np.float64)
    for k in range(divisor):
        total += np.array([dict_info[t - k * time_delta_hi] for t in stamps], dtype=np.float64)
    return dict(zip(stamps, total))


def rescale_time_resolution_of_regular_cumulative_data(dict_data, start_data, end_data, time_delta_data,
                                                       time_delta_res, start_simu, end_simu, time_delta_simu):
    """timeframe.py:211-233: onto the common grid if the data are coarser than it, then onto the simulation stamps."""
    fine = dict_data
    if time_delta_data > time_delta_res:
        fine = increase_time_resolution_of_regular_cumulative_data(dict_data, start_data, end_data,
                                                                   time_delta_data, time_delta_res)
    return decrease_time_resolution_of_regular_cumulative_data(fine, start_simu, end_simu, time_delta_simu,
                                                               time_delta_res)


def increase_time_resolution_of_irregular_mean_data(dict_info, time_delta_lo, time_delta_hi):
    """Mean values replicated backwards onto the fine grid, over the span back to the previous stamp -- one standard
    interval for the first stamp and after a gap of 1.5 intervals or more (timeframe.py:236-271)."""
    stamps = list(dict_info)
    lo = _isec(time_delta_lo)
    secs = np.array([_sec(t) for t in stamps], dtype=np.int64)
    span = np.diff(secs, prepend=secs[0] - lo) if len(secs) else secs
    span = np.where(span >= 1.5 * lo, lo, span)
    out = dict()
    for t, width in zip(stamps, span):
        divisor = _divisor(timedelta(seconds=int(width)), time_delta_hi, "Increase")
        try:
            value = float(dict_info[t])
        except ValueError:          # a string that is not a number: no data for this stamp
            value = float('nan')
        for k in range(divisor):
            if out.get(t - k * time_delta_hi):
                raise Exception("Increase Resolution: Overwriting already existing data for datetime.")
            out[t - k * time_delta_hi] = value
    return out


def decrease_time_resolution_of_irregular_mean_data(dict_info, dt_start, dt_end, time_delta_hi, time_delta_lo):
    """Arithmetic mean of the sub-steps ending at each low-resolution stamp, NaN if one is missing
    (timeframe.py:274-301)."""
    divisor = _divisor(time_delta_lo, time_delta_hi, "Decrease")
    out = OrderedDict()
    for t in TimeFrame._series(dt_start, dt_end, time_delta_lo):
        try:
            total = 0.0
            for k in range(divisor):
                total += dict_info[t - k * time_delta_hi]
            out[t] = total / divisor
        except (KeyError, TypeError):
            out[t] = float('nan')
    return out


def rescale_time_resolution_of_irregular_mean_data(dict_data, start_data, end_data, time_delta_lo, time_delta_hi):
    """timeframe.py:304-309."""
    fine = increase_time_resolution_of_irregular_mean_data(dict_data, time_delta_lo, time_delta_hi)
    return decrease_time_resolution_of_irregular_mean_data(fine, start_data, end_data, time_delta_hi, time_delta_lo)
