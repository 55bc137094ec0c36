"""ctypes front-end of the parity oracle (oracle/smart_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product (smartpy_amd) never does.  Builds the shared object on first use with oracle/Makefile.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libsmart_oracle.so")

POW_LIBM, POW_MUL = 0, 1
SUM_NUMPY, SUM_SEQ, SUM_GPU = 0, 1, 2
REPORT_SUMMARY, REPORT_RAW = 1, 2
NVAR = 19

_lib = None
_dp = ctypes.POINTER(ctypes.c_double)


def build(force=False):
    """Build libsmart_oracle.so with oracle/Makefile when it is missing or older than its source.  Atomic: the compiler
    writes a name of this process's own and the finished file is renamed onto the library (os.replace), so that a
    process that loads the library while another one builds it -- the ranks of a multi-GPU bench on a freshly pushed
    tree, where the mtimes are arbitrary -- sees either the old file or the new one, never half of one.  A library
    that is current is left alone (no `make -B`)."""
    src = os.path.join(_HERE, "smart_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        tmp = "libsmart_oracle.%d.tmp.so" % os.getpid()
        try:
            subprocess.check_call(["make", "-s", "-C", _HERE, "-B", tmp])
            os.replace(os.path.join(_HERE, tmp), _SO)
        finally:
            if os.path.exists(os.path.join(_HERE, tmp)):
                os.remove(os.path.join(_HERE, tmp))
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        L.smart_oracle_one_step.restype = None
        L.smart_oracle_one_step.argtypes = [ctypes.c_double] * 4 + [_dp, _dp, _dp, ctypes.c_int]
        L.smart_oracle_all_steps.restype = ctypes.c_int
        L.smart_oracle_all_steps.argtypes = [ctypes.c_double, ctypes.c_double, ctypes.c_long, _dp, _dp, _dp, _dp,
                                             ctypes.c_int, ctypes.c_long, ctypes.c_int, ctypes.c_int,
                                             _dp, _dp, _dp, _dp]
        L.smart_oracle_initial.restype = None
        L.smart_oracle_initial.argtypes = [ctypes.c_double, _dp, _dp, _dp]
        L.smart_oracle_run.restype = ctypes.c_int
        L.smart_oracle_run.argtypes = [ctypes.c_double, ctypes.c_double, ctypes.c_long, ctypes.c_long, _dp, _dp,
                                       _dp, _dp, ctypes.c_int, ctypes.c_long, ctypes.c_int, ctypes.c_int,
                                       _dp, _dp, _dp]
        L.smart_oracle_run_batch.restype = ctypes.c_int
        L.smart_oracle_run_batch.argtypes = [ctypes.c_long, ctypes.c_double, ctypes.c_double, ctypes.c_long,
                                             ctypes.c_long, _dp, _dp, _dp, _dp, ctypes.c_int, ctypes.c_long,
                                             ctypes.c_int, ctypes.c_int, _dp, _dp, _dp, ctypes.c_int]
        L.smart_oracle_max_threads.restype = ctypes.c_int
        _lib = L
    return _lib


def _c(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def _extra_vec(extra):
    """{'aar','r-o_ratio','r-o_split'} -> 7 doubles, or None (structure.py:100-112)."""
    if not extra:
        return None
    return np.array([extra['aar'], extra['r-o_ratio']] + list(extra['r-o_split']), dtype=np.float64)


def one_step(area, dt, rain, peva, params, states, pow_mode=POW_LIBM):
    """structure.py:200 run_one_step -> 19-vector."""
    p, pp = _c(params)
    s, sp = _c(states)
    out = np.empty(NVAR)
    lib().smart_oracle_one_step(area, dt, rain, peva, pp, sp, out.ctypes.data_as(_dp), pow_mode)
    return out


def all_steps(area, dt, length, rain, peva, params, initial, report_type, gap,
              pow_mode=POW_LIBM, sum_mode=SUM_NUMPY, want_storage=False):
    """structure.py:149 run_all_steps -> (discharge, gw, final[, storage])."""
    r, rp = _c(rain)
    e, ep = _c(peva)
    p, pp = _c(params)
    i, ip = _c(initial)
    R = length // gap if report_type == REPORT_SUMMARY else -(-length // gap)
    dis = np.empty(max(R, 0))
    gw = ctypes.c_double(0.0)
    fin = np.empty(NVAR)
    sto = np.empty((length + 1, NVAR)) if want_storage else None
    rc = lib().smart_oracle_all_steps(area, dt, length, rp, ep, pp, ip, report_type, gap, pow_mode, sum_mode,
                                      dis.ctypes.data_as(_dp), ctypes.byref(gw), fin.ctypes.data_as(_dp),
                                      sto.ctypes.data_as(_dp) if want_storage else None)
    if rc:
        raise Exception("smart_oracle_all_steps failed with code %d" % rc)
    return (dis, gw.value, fin, sto) if want_storage else (dis, gw.value, fin)


def initial(area, params, extra):
    p, pp = _c(params)
    x = _extra_vec(extra)
    out = np.empty(NVAR)
    lib().smart_oracle_initial(area, pp, x.ctypes.data_as(_dp) if x is not None else None, out.ctypes.data_as(_dp))
    return out


def run(area, dt, n_steps, n_warm, rain, peva, params, extra, report_type, gap,
        pow_mode=POW_LIBM, sum_mode=SUM_NUMPY):
    """structure.py:30 run() for one sample -> (discharge, gw, final)."""
    r, rp = _c(rain)
    e, ep = _c(peva)
    p, pp = _c(params)
    x = _extra_vec(extra)
    R = n_steps // gap if report_type == REPORT_SUMMARY else -(-n_steps // gap)
    dis = np.empty(R)
    gw = ctypes.c_double(0.0)
    fin = np.empty(NVAR)
    rc = lib().smart_oracle_run(area, dt, n_steps, n_warm, rp, ep, pp,
                                x.ctypes.data_as(_dp) if x is not None else None,
                                report_type, gap, pow_mode, sum_mode,
                                dis.ctypes.data_as(_dp), ctypes.byref(gw), fin.ctypes.data_as(_dp))
    if rc:
        raise Exception("smart_oracle_run failed with code %d" % rc)
    return dis, gw.value, fin


def run_batch(area, dt, n_steps, n_warm, rain, peva, params, extra, report_type, gap,
              pow_mode=POW_LIBM, sum_mode=SUM_NUMPY, want_discharge=True, want_final=False, n_threads=0):
    """N samples at once (OpenMP) -> (discharge[N,R] | None, gw[N], final[N,19] | None)."""
    r, rp = _c(rain)
    e, ep = _c(peva)
    p, pp = _c(params)
    assert p.ndim == 2 and p.shape[1] == 10
    N = p.shape[0]
    x = _extra_vec(extra)
    R = n_steps // gap if report_type == REPORT_SUMMARY else -(-n_steps // gap)
    dis = np.empty((N, R)) if want_discharge else None
    gw = np.empty(N)
    fin = np.empty((N, NVAR)) if want_final else None
    rc = lib().smart_oracle_run_batch(N, area, dt, n_steps, n_warm, rp, ep, pp,
                                      x.ctypes.data_as(_dp) if x is not None else None,
                                      report_type, gap, pow_mode, sum_mode,
                                      dis.ctypes.data_as(_dp) if want_discharge else None,
                                      gw.ctypes.data_as(_dp),
                                      fin.ctypes.data_as(_dp) if want_final else None, n_threads)
    if rc:
        raise Exception("smart_oracle_run_batch failed with code %d" % rc)
    return dis, gw, fin


def max_threads():
    return lib().smart_oracle_max_threads()
