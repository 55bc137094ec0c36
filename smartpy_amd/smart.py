"""The model facade (counterpart of smartpy/smart.py): same constructor, attributes and methods, GPU underneath.

`SMART.simulate(param)` runs one parameter set like the reference (smart.py:154-210).  The MI355X-native entry is
`SMART.simulate_ensemble(matrix)`: every row of an [N, 10] parameter matrix in one launch, optionally with the
objective functions fused, which is what the Monte-Carlo classes call.
"""
from os import makedirs, sep

import numpy as np

from .timeframe import TimeFrame
from .parameters import Parameters
from .inout import get_rain_series_simu, get_peva_series_simu, get_discharge_series, write_flow_file_from_nds
from . import structure
from . import engine


class SMART(object):
    """SMART is the core object to set up and use to run an experiment (smart.py:29)."""

    def __init__(self, catchment, catchment_area_m2, start, end,
                 time_delta_simu, time_delta_save, warm_up_days,
                 in_format, out_format, root,
                 gauged_area_m2=None):
        self._init_common(catchment, catchment_area_m2, start, end, time_delta_simu, time_delta_save, warm_up_days,
                          in_format, out_format, root)
        makedirs(self.out_f, exist_ok=True)      # every rank of a multi-GPU run gets here at the same time
        ext = '.nc' if self.in_fmt == 'netcdf' else ''
        base = ''.join([self.in_f, self.catchment])
        # forcing on the simulation axis, observations on the report axis (smart.py:130-143)
        self.nd_rain = get_rain_series_simu(base + '.rain' + ext, self.in_fmt, self.timeseries[1],
                                            self.timeseries[-1], self.delta_simu)
        self.nd_peva = get_peva_series_simu(base + '.peva' + ext, self.in_fmt, self.timeseries[1],
                                            self.timeseries[-1], self.delta_simu)
        self.nd_flow = get_discharge_series(base + '.flow' + ext, self.in_fmt, self.timeseries_report[1],
                                            self.timeseries_report[-1], catchment_area_m2, gauged_area_m2) \
            if gauged_area_m2 else None

    @classmethod
    def from_arrays(cls, catchment_area_m2, start, end, time_delta_simu, time_delta_save, warm_up_days,
                    nd_rain, nd_peva, nd_flow=None, catchment='catchment', out_format='csv', root=None):
        """Build a model from in-memory series instead of files: nd_rain / nd_peva per simulation step (length T),
        nd_flow per report step (length R, NaN = missing) or None."""
        self = cls.__new__(cls)
        self._init_common(catchment, catchment_area_m2, start, end, time_delta_simu, time_delta_save, warm_up_days,
                          'csv', out_format, root if root is not None else '.')
        self.nd_rain = np.ascontiguousarray(nd_rain, dtype=np.float64)
        self.nd_peva = np.ascontiguousarray(nd_peva, dtype=np.float64)
        self.nd_flow = None if nd_flow is None else np.ascontiguousarray(nd_flow, dtype=np.float64)
        T, R = len(self.timeseries) - 1, len(self.timeseries_report) - 1
        if len(self.nd_rain) != T or len(self.nd_peva) != T or (self.nd_flow is not None and len(self.nd_flow) != R):
            raise Exception("from_arrays: expected {} forcing values and {} observations.".format(T, R))
        return self

    def _init_common(self, catchment, catchment_area_m2, start, end, time_delta_simu, time_delta_save,
                     warm_up_days, in_format, out_format, root):
        """Everything of the constructor that does not read files: identity, folders, the two time axes, and the
        empty result slots (attribute names as in smart.py:109-152, which user scripts read)."""
        self.catchment, self.area = catchment, catchment_area_m2
        self.in_fmt, self.out_fmt, self.root_f = in_format, out_format, root
        self.in_f, self.out_f = (sep.join([root, side, catchment, sep]) for side in ('in', 'out'))
        # simulation axis (T + 1 stamps, one leading initial-condition stamp) and report axis (R + 1)
        self.start, self.end = start, end
        self.delta_simu, self.delta_save = time_delta_simu, time_delta_save
        self.timeframe = TimeFrame(start, end, time_delta_simu, time_delta_save)
        self.timeseries = self.timeframe.get_series_simu()
        self.timeseries_report = self.timeframe.get_series_save()
        self.warm_up = warm_up_days
        self.parameters = Parameters()
        self.extra = None                    # educated guess of the initial reservoirs, set by the user (smart.py:145)
        self.outputs = self.nd_discharge = self.gw_contribution = None
        self._device_cache = None
        self._single = {}                    # engine.SingleRun per (report, device, extra ...): simulate()

    # ------------------------------------------------------------------------------------------------------
    def simulate(self, param, report='summary'):
        """One parameter set (dict with the ten names) -> (discharge ndarray [R], gw float) (smart.py:154-210).

        The forcing series, the output buffers and the launch plan stay on the device between calls (engine.SingleRun,
        one per report type): a calibration loop over simulate() -- the reference's per-sample protocol,
        montecarlo.py:179-186 -- uploads ten numbers per call and nothing else."""
        nd_parameters = np.array([param[name] for name in self.parameters.names])
        report_type = engine.report_code(report)                    # raises on an unknown report type (structure.py:69-70)
        T = len(self.timeseries) - 1
        delta_sec = self.delta_simu.total_seconds()
        gap = T // (len(self.timeseries_report) - 1)
        n_warm = structure.warm_up_length(self.warm_up, delta_sec, T) if self.warm_up != 0 else 0
        if report_type == engine.REPORT_SUMMARY and (T % gap or n_warm % gap):
            bad = T if T % gap else n_warm      # what np.reshape raises in the reference (structure.py:190)
            raise ValueError("cannot reshape array of size {} into shape ({})".format(bad, gap))
        device = engine.default_device()
        extra = tuple(engine.extra_vector(self.extra)) if self.extra else None
        forcing = self._device_forcing(device)          # (compares the series with what is on the device: see there)
        # what a kept run was made for; `forcing` is a new tensor whenever the series changed, so its identity counts --
        # and so do the lengths a user script may change between calls by replacing timeseries / timeseries_report /
        # delta_simu (round 5's key left them out: the old gap, warm-up and output length were reused silently; the
        # reference's structure.run recomputes them on every call)
        key = (report, str(device), extra, float(self.area), self.warm_up, id(forcing), T, gap, n_warm, delta_sec)
        run = self._single.get(key)
        if run is None:
            if len(self._single) >= 4:
                self._single.clear()
            run = self._single[key] = engine.SingleRun(forcing, float(self.area), delta_sec, n_warm, gap, report=report,
                                                       extra=self.extra if self.extra else None, device=device)
        self.outputs = run.run(nd_parameters)
        self.nd_discharge = self.outputs[0]
        self.gw_contribution = self.outputs[1]
        return self.outputs

    def _device_forcing(self, device):
        """The [T, 2] forcing and the observations on the device, uploaded once per (device, series).  `nd_rain`,
        `nd_peva` and `nd_flow` are plain attributes that user scripts may replace OR write into: the library keeps a
        host copy of what it uploaded and compares on every call (like the smartcpp hook does: ~0.1 ms for ten years of
        hourly values) -- a series that differs is uploaded again, and whatever was prepared for the old one is dropped."""
        import torch
        device = torch.device(device)
        T = len(self.timeseries) - 1
        rain = np.asarray(self.nd_rain, dtype=np.float64)[:T]
        peva = np.asarray(self.nd_peva, dtype=np.float64)[:T]
        flow = None if self.nd_flow is None else np.asarray(self.nd_flow, dtype=np.float64)
        c = self._device_cache

        def same(a, b):         # (NaN = missing observation: equal to itself here)
            return (a is None and b is None) or (a is not None and b is not None and a.shape == b.shape and
                                                 np.array_equal(a, b, equal_nan=True))
        if c is None or c[0] != device or not (same(c[3], rain) and same(c[4], peva) and same(c[5], flow)):
            forcing = engine.as_device(np.stack([rain, peva], axis=1), device)
            obs = engine.as_device(flow, device) if flow is not None else None
            self._device_cache = (device, forcing, obs, rain.copy(), peva.copy(), None if flow is None else flow.copy())
            self._single.clear()
        return self._device_cache[1]

    def simulate_ensemble(self, parameters, report='summary', objective_functions=False, gw_constraint=None,
                          save_discharge=True, math_mode='fast', device=None):
        """Every row of `parameters` ([N, 10] ndarray or device tensor, columns in self.parameters.names order)
        in one launch.  Returns an engine.EnsembleResult with device tensors: .discharge [N, R] (a view of the
        sample-minor buffer), .gw [N], .objfn [N, 8] when objective_functions (needs observations)."""
        import torch
        device = torch.device(device) if device is not None else engine.default_device()
        delta_sec = self.delta_simu.total_seconds()
        T = len(self.timeseries) - 1
        n_warm = structure.warm_up_length(self.warm_up, delta_sec, T) if self.warm_up != 0 else 0
        gap = T // (len(self.timeseries_report) - 1)
        forcing = self._device_forcing(device)
        obs = self._device_cache[2]
        if objective_functions and obs is None:
            raise Exception("The observation array does not exist. Please make sure that a value is assigned "
                            "to the gauged_area_m2 attribute of your SMART class instance.")
        return engine.run_ensemble(parameters, forcing, float(self.area), delta_sec, n_warm, gap, report=report,
                                   extra=self.extra if self.extra else None, obs=obs if objective_functions else None,
                                   gw_obs=gw_constraint if objective_functions else None, math_mode=math_mode,
                                   want_discharge=save_discharge, want_objfn=bool(objective_functions),
                                   device=device)

    # ------------------------------------------------------------------------------------------------------
    # the two flow series a model can write / hand out: attribute holding it, file suffix, and what to tell the
    # user when it is not there yet (the reference's messages, smart.py:212-277)
    _SERIES = {
        'modelled': ('nd_discharge', '.mod.flow',
                     "The modelled flow output file cannot be written. Please make sure to call the "
                     "simulate method of your SMART instance before writing this output file."),
        'observed': ('nd_flow', '.obs.flow',
                     "The observed flow output file cannot be written. Please make sure that a value is "
                     "assigned to the gauged_area_m2 attribute of the SMART class instance."),
    }

    def write_output_files(self, which='both', parallel=False):
        """Write `<out>/<catchment>.mod.flow` and / or `.obs.flow` on the report time axis (smart.py:212-255);
        which = 'modelled' | 'observed' | 'both' (modelled first)."""
        for kind in (('modelled', 'observed') if which == 'both' else (which,)):
            if kind not in self._SERIES:
                continue
            attribute, suffix, problem = self._SERIES[kind]
            series = getattr(self, attribute)
            if series is None:
                raise Exception(problem)
            write_flow_file_from_nds(self.timeseries_report[1:], series, self.out_f + self.catchment + suffix,
                                     out_file_format=self.out_fmt, parallel=parallel)

    def _series_or_raise(self, attribute, problem):
        series = getattr(self, attribute)
        if series is None:
            raise Exception(problem)
        return series

    def get_simulation_array(self):
        return self._series_or_raise(
            'nd_discharge', "The simulation array cannot be retrieved. Please make sure to call the simulate "
                            "method of your SMART instance before requesting this output array.")

    def get_evaluation_array(self):
        return self._series_or_raise(
            'nd_flow', "The observation array does not exist. Please make sure that a value is assigned "
                       "to the gauged_area_m2 attribute of your SMART class instance.")
