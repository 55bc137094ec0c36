"""File formats either side of the hot path (counterpart of smartpy/inout.py).

Readers for the `.rain` / `.peva` / `.flow` / `.sttngs` inputs and writers for the `.mod.flow` / `.obs.flow`
outputs, in the reference's formats (inout.py:35-310).  Series are returned as numpy arrays on the
simulation / report axes, ready to be uploaded to the GPU; the resampling itself is in timeframe.py.
NetCDF needs the optional netCDF4 package, exactly as in the reference (inout.py:25-28).
"""
import argparse
from collections import OrderedDict
from csv import DictReader, writer
from datetime import datetime, timedelta

import numpy as np

try:
    from netCDF4 import Dataset
except ImportError:
    Dataset = None

from .timeframe import TimeFrame, check_interval_in_list, get_required_resolution, resample_cumulative, \
    resample_irregular_mean
from .version import __version__

_NO_NETCDF_IN = "The use of 'netcdf' as the input file format requires the package 'netCDF4', " \
                "please install it and retry, or choose another file format."
_NO_NETCDF_OUT = "The use of 'netcdf' as the output file format requires the package 'netCDF4', " \
                 "please install it and retry, or choose another file format."
_DT = "%Y-%m-%d %H:%M:%S"


def _stamp(text):
    """'%Y-%m-%d %H:%M:%S' -> datetime.  datetime.fromisoformat is ten times faster than strptime (a ten-year daily
    file has 3,653 stamps, three files per catchment); anything that is not exactly of that shape goes through
    strptime, which accepts or rejects it as the reference does."""
    if len(text) == 19 and text[10] == ' ':
        try:
            return datetime.fromisoformat(text)
        except ValueError:
            pass
    return datetime.strptime(text, _DT)


# ----------------------------------------------------------------------------------------------------------
# forcing
# ----------------------------------------------------------------------------------------------------------
def _read_regular_series(file_location, file_format, variable):
    """-> (stamps list, values float64 array, start, end, interval); regular spacing enforced (inout.py:192-231)."""
    if file_format == 'netcdf':
        if not Dataset:
            raise Exception(_NO_NETCDF_IN)
        try:
            with Dataset(file_location, 'r') as f:
                f.set_auto_mask(False)
                try:
                    stamps = [datetime(1970, 1, 1) + timedelta(seconds=float(t)) for t in f.variables['DateTime'][:]]
                    values = np.asarray(f.variables[variable][:], dtype=np.float64)
                except KeyError:
                    raise Exception('Variable {} or {} does not exist in {}.'.format('DateTime', variable,
                                                                                     file_location))
        except IOError:
            raise Exception('File {} could not be found.'.format(file_location))
    else:
        try:
            with open(file_location, 'r', encoding='utf8') as f:
                stamps, values = [], []
                try:
                    for row in DictReader(f):
                        stamps.append(_stamp(row['DateTime']))
                        values.append(np.float64(row[variable]))
                except KeyError:
                    raise Exception('Field {} or {} does not exist in {}.'.format('DateTime', variable,
                                                                                  file_location))
        except IOError:
            raise Exception('File {} could not be found.'.format(file_location))
        values = np.asarray(values, dtype=np.float64)
    start, end, interval = check_interval_in_list(stamps, file_location)
    return stamps, values, start, end, interval


def _forcing_series(file_location, file_format, variable, label, start_simu, end_simu, time_delta_simu):
    stamps, values, start_data, end_data, delta_data = _read_regular_series(file_location, file_format, variable)
    if (start_data - delta_data + time_delta_simu <= start_simu) and (end_simu <= end_data):   # inout.py:38,51
        delta_res = get_required_resolution(start_data, start_simu, delta_data, time_delta_simu)
        return resample_cumulative(values, start_data, delta_data, delta_res, start_simu, end_simu,
                                   time_delta_simu)
    raise Exception('{} data not sufficient for simulation.'.format(label))


def get_rain_series_simu(file_location, file_format, start_simu, end_simu, time_delta_simu):
    """Rainfall per simulation step (mm / step) for the stamps start_simu .. end_simu (inout.py:35-45)."""
    return _forcing_series(file_location, file_format, 'rain', 'Rain', start_simu, end_simu, time_delta_simu)


def get_peva_series_simu(file_location, file_format, start_simu, end_simu, time_delta_simu):
    """Potential evapotranspiration per simulation step (inout.py:48-58)."""
    return _forcing_series(file_location, file_format, 'peva', 'PEva', start_simu, end_simu, time_delta_simu)


# ----------------------------------------------------------------------------------------------------------
# observations
# ----------------------------------------------------------------------------------------------------------
def _read_flow_arrays(file_location, file_format):
    """-> (stamps, values) with missing entries dropped: '' / -99 in CSV (inout.py:234-254), NaN in NetCDF
    (inout.py:257-274)."""
    if file_format == 'netcdf':
        if not Dataset:
            raise Exception(_NO_NETCDF_IN)
        try:
            with Dataset(file_location, 'r') as f:
                try:
                    times = f.variables['DateTime'][:]
                    flows = np.asarray(f.variables['flow'][:], dtype=np.float64)
                except KeyError:
                    raise Exception('Variable {} or {} does not exist in {}.'.format('DateTime', 'flow',
                                                                                     file_location))
                keep = ~np.isnan(flows)
                stamps = [datetime(1970, 1, 1) + timedelta(seconds=float(t)) for t, k in zip(times, keep) if k]
                return stamps, flows[keep]
        except IOError:
            raise Exception('File {} could not be found.'.format(file_location))
    try:
        with open(file_location, 'r', encoding='utf8') as f:
            stamps, values = [], []
            try:
                for row in DictReader(f):
                    try:
                        if row['flow'] != '':
                            v = np.float64(row['flow'])
                            if v != -99.0:
                                stamps.append(_stamp(row['DateTime']))
                                values.append(v)
                    except ValueError:
                        raise Exception('Field {} in {} cannot be converted to float '
                                        'at {}.'.format('flow', file_location, row['DateTime']))
            except KeyError:
                raise Exception('Field {} or {} does not exist in {}.'.format('DateTime', 'flow', file_location))
        return stamps, np.asarray(values, dtype=np.float64)
    except IOError:
        raise Exception('File {} could not be found.'.format(file_location))


def get_discharge_series(file_location, file_format, start_report, end_report, catchment_area, gauged_area):
    """Observed discharge per report stamp, rescaled by catchment / gauged area, NaN where missing
    (inout.py:61-78; the daily-mean / hourly-replication assumption of :77-78 is the reference's)."""
    stamps, values = _read_flow_arrays(file_location, file_format)
    scaling_factor = catchment_area / gauged_area
    first_day = (start_report - timedelta(days=2)).date()
    last_day = (end_report + timedelta(days=1)).date()
    keep = [i for i, dt in enumerate(stamps) if first_day <= dt.date() <= last_day]
    sel_stamps = [stamps[i] for i in keep]
    sel_values = values[keep] * scaling_factor
    return resample_irregular_mean(sel_stamps, sel_values, start_report, end_report,
                                   timedelta(days=1), timedelta(hours=1))


# ----------------------------------------------------------------------------------------------------------
# settings
# ----------------------------------------------------------------------------------------------------------
def read_simulation_settings_file(file_location):
    """ARGUMENT,VALUE CSV -> dict (inout.py:177-189)."""
    args = dict()
    try:
        with open(file_location, 'r', encoding='utf8') as f:
            for row in DictReader(f):
                args[row['ARGUMENT']] = row['VALUE']
    except KeyError:
        raise Exception("There is no 'ARGUMENT' or 'VALUE' column in {}.".format(file_location))
    except IOError:
        raise Exception("There is no simulation file at {}.".format(file_location))
    return args


def get_dict_simulation_settings(file_location):
    """-> (c_area m2, g_area m2, start, end, delta_simu, delta_report, warm_up days, gw_constraint | None)
    with the reference's keys, units, defaults and messages (inout.py:81-140)."""
    a = read_simulation_settings_file(file_location)

    def need(key, convert, name, what):
        try:
            return convert(a[key])
        except KeyError:
            raise Exception('Setting {} is missing from simulation file.'.format(name))
        except ValueError:
            raise Exception('Setting {} could not be converted to {}.'.format(name, what))

    c_area = need('catchment_area_km2', lambda v: float(v) * 1e6, 'CATCHMENT AREA', 'a float')
    try:
        g_area = float(a['gauged_area_km2']) * 1e6
    except KeyError:
        g_area = c_area
    except ValueError:
        raise Exception('Setting GAUGED AREA could not be converted to a float.')
    fmt = '%d/%m/%Y %H:%M:%S'
    start = need('start_datetime', lambda v: datetime.strptime(v, fmt), 'START',
                 'a datetime [format required: DD/MM/YYYY HH:MM:SS]')
    end = need('end_datetime', lambda v: datetime.strptime(v, fmt), 'END',
               'a datetime [format required: DD/MM/YYYY HH:MM:SS]')
    delta_simu = need('simu_timedelta_min', lambda v: timedelta(minutes=int(v)), 'DELTA SIMU', 'an integer/timedelta')
    delta_report = need('report_timedelta_min', lambda v: timedelta(minutes=int(v)), 'DELTA REPORT',
                        'an integer/timedelta')
    warm_up = need('warm_up_days', int, 'WARM UP DURATION', 'an integer')
    try:
        gw_constraint = float(a['gw_constraint'])
    except KeyError:
        gw_constraint = None
    except ValueError:
        raise Exception('Setting GROUNDWATER CONSTRAINT could not be converted to a float.')
    return c_area, g_area, start, end, delta_simu, delta_report, warm_up, gw_constraint


# ----------------------------------------------------------------------------------------------------------
# discharge writers
# ----------------------------------------------------------------------------------------------------------
def write_flow_file_from_nds(series_report, discharge, the_file, out_file_format, parallel=False):
    """inout.py:277-288."""
    if out_file_format == 'netcdf':
        if not Dataset:
            raise Exception(_NO_NETCDF_OUT)
        write_flow_netcdf_file_from_nds(series_report, discharge, the_file, parallel=parallel)
    elif out_file_format == 'csv':
        write_flow_csv_file_from_nds(series_report, discharge, the_file)
    else:
        raise Exception("The output format type \'{}\' cannot be written by SMARTpy, "
                        "choose from: \'csv\', \'netcdf\'.".format(out_file_format))


def write_flow_csv_file_from_nds(series_report, discharge, csv_file):
    """DateTime,flow rows, values as '%e', csv.writer line endings (inout.py:291-296)."""
    with open(csv_file, 'w', newline='', encoding='utf8') as f:
        w = writer(f, delimiter=',')
        w.writerow(['DateTime', 'flow'])
        w.writerows([dt, '%e' % val] for dt, val in zip(series_report, discharge))


def write_flow_netcdf_file_from_nds(series_report, discharge, netcdf_file, parallel):
    """DateTime (f64 epoch seconds) + flow (f32) (inout.py:299-310)."""
    with Dataset(netcdf_file + '.nc', 'w', format='NETCDF4', parallel=parallel) as f:
        f.description = "Discharge file generated with SMARTpy v{}.".format(__version__)
        f.createDimension('DateTime', len(series_report))
        t = f.createVariable("DateTime", np.float64, ('DateTime',))
        t.units = 'seconds since 1970-01-01 00:00:00.0'
        f.createVariable('flow', np.float32, ('DateTime',))
        f.variables['DateTime'][0:len(series_report)] = \
            (np.asarray(series_report, dtype='datetime64[us]') - np.datetime64('1970-01-01T00:00:00')) / \
            np.timedelta64(1, 's')
        f.variables['flow'][0:len(series_report)] = discharge


# ----------------------------------------------------------------------------------------------------------
# The reference's dictionary-returning readers (inout.py:35-78, 143-274, 313-322), for scripts that call them
# directly: same names, arguments, results and error texts, on top of the array readers above.
# ----------------------------------------------------------------------------------------------------------
def read_csv_time_series_with_delta_check(csv_file, key_header, val_header):
    """-> (dict stamp -> value, first stamp, last stamp, interval); regular spacing enforced (inout.py:192-209)."""
    try:
        with open(csv_file, 'r', encoding='utf8') as f:
            data = dict()
            stamps = []
            try:
                for row in DictReader(f):
                    stamp = _stamp(row[key_header])
                    data[stamp] = np.float64(row[val_header])
                    stamps.append(stamp)
            except KeyError:
                raise Exception('Field {} or {} does not exist in {}.'.format(key_header, val_header, csv_file))
        start, end, interval = check_interval_in_list(stamps, csv_file)
        return data, start, end, interval
    except IOError:
        raise Exception('File {} could not be found.'.format(csv_file))


def read_netcdf_time_series_with_delta_check(netcdf_file, key_variable, val_variable):
    """inout.py:212-231."""
    if not Dataset:
        raise Exception(_NO_NETCDF_IN)
    try:
        with Dataset(netcdf_file, 'r') as f:
            f.set_auto_mask(False)
            try:
                stamps = [datetime(1970, 1, 1) + timedelta(seconds=float(t)) for t in f.variables[key_variable][:]]
                data = dict(zip(stamps, f.variables[val_variable][:]))
            except KeyError:
                raise Exception('Variable {} or {} does not exist in {}.'.format(key_variable, val_variable,
                                                                                 netcdf_file))
        start, end, interval = check_interval_in_list(stamps, netcdf_file)
        return data, start, end, interval
    except IOError:
        raise Exception('File {} could not be found.'.format(netcdf_file))


def read_csv_time_series_with_missing_check(csv_file, key_header, val_header):
    """-> OrderedDict stamp -> value without the missing entries ('' or -99) (inout.py:234-253)."""
    try:
        with open(csv_file, 'r', encoding='utf8') as f:
            data = OrderedDict()
            try:
                for row in DictReader(f):
                    try:
                        if row[val_header] != '' and np.float64(row[val_header]) != -99.0:
                            data[_stamp(row[key_header])] = np.float64(row[val_header])
                    except ValueError:
                        raise Exception('Field {} in {} cannot be converted to float '
                                        'at {}.'.format(val_header, csv_file, row[key_header]))
            except KeyError:
                raise Exception('Field {} or {} does not exist in {}.'.format(key_header, val_header, csv_file))
        return data
    except IOError:
        raise Exception('File {} could not be found.'.format(csv_file))


def read_netcdf_time_series_with_missing_check(netcdf_file, key_variable, val_variable):
    """inout.py:256-274."""
    if not Dataset:
        raise Exception(_NO_NETCDF_IN)
    try:
        with Dataset(netcdf_file, 'r') as f:
            data = OrderedDict()
            try:
                stamps = [datetime(1970, 1, 1) + timedelta(seconds=float(t)) for t in f.variables[key_variable][:]]
                for stamp, flow in zip(stamps, f.variables[val_variable][:]):
                    if not np.isnan(flow):
                        data[stamp] = flow
            except KeyError:
                raise Exception('Variable {} or {} does not exist in {}.'.format(key_variable, val_variable,
                                                                                 netcdf_file))
        return data
    except IOError:
        raise Exception('File {} could not be found.'.format(netcdf_file))


def _read_with(reader_csv, reader_netcdf, file_location, file_format, variable):
    if file_format == 'netcdf':
        if not Dataset:
            raise Exception(_NO_NETCDF_IN)
        return reader_netcdf(file_location, key_variable='DateTime', val_variable=variable)
    return reader_csv(file_location, key_header='DateTime', val_header=variable)


def read_rain_file(file_location, file_format):
    """inout.py:143-151."""
    return _read_with(read_csv_time_series_with_delta_check, read_netcdf_time_series_with_delta_check,
                      file_location, file_format, 'rain')


def read_peva_file(file_location, file_format):
    """inout.py:154-162."""
    return _read_with(read_csv_time_series_with_delta_check, read_netcdf_time_series_with_delta_check,
                      file_location, file_format, 'peva')


def read_flow_file(file_location, file_format):
    """inout.py:165-174."""
    return _read_with(read_csv_time_series_with_missing_check, read_netcdf_time_series_with_missing_check,
                      file_location, file_format, 'flow')


def _keyed(first, last, gap, values):
    return OrderedDict(zip(TimeFrame._series(first, last, gap), values.tolist()))


def get_dict_rain_series_simu(file_location, file_format, start_simu, end_simu, time_delta_simu):
    """inout.py:35-45: rain per simulation step, keyed by the simulation stamps."""
    return _keyed(start_simu, end_simu, time_delta_simu,
                  get_rain_series_simu(file_location, file_format, start_simu, end_simu, time_delta_simu))


def get_dict_peva_series_simu(file_location, file_format, start_simu, end_simu, time_delta_simu):
    """inout.py:48-58."""
    return _keyed(start_simu, end_simu, time_delta_simu,
                  get_peva_series_simu(file_location, file_format, start_simu, end_simu, time_delta_simu))


def get_dict_discharge_series(file_location, file_format, start_report, end_report, catchment_area, gauged_area):
    """inout.py:61-78: rescaled observed discharge per daily report stamp, NaN where missing."""
    return _keyed(start_report, end_report, timedelta(days=1),
                  get_discharge_series(file_location, file_format, start_report, end_report, catchment_area,
                                       gauged_area))


def valid_file_format(fmt):
    """argparse type (inout.py:313-322)."""
    if fmt.lower() == "netcdf":
        if Dataset:
            return "netcdf"
        raise argparse.ArgumentTypeError("NetCDF4 module is not installed, please choose another file format.")
    if fmt.lower() == "csv":
        return "csv"
    raise argparse.ArgumentTypeError("File format not recognised: '{0}'.".format(fmt))
