"""ctypes binding of libsmart_amd.so (the C ABI of include/smart_amd.h).

The library is the product: there is no CPU fallback.  If the shared object is missing (not built) or
cannot be loaded, importing the engine fails loudly with instructions, and every compute entry point
returns SMART_E_NO_DEVICE when no HIP device is visible.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SMART_AMD_LIB selects another build of the same ABI (kernel tuning experiments: tools/ab_variants.sh)
LIB_PATH = os.environ.get('SMART_AMD_LIB') or os.path.join(_HERE, 'csrc', 'libsmart_amd.so')

REPORT_SUMMARY, REPORT_RAW = 1, 2
MATH_LITERAL, MATH_FAST = 0, 1
ABI_VERSION = 7
PLAN_VALID = 0x100
PLAN_CLASS_BITS = {0: 0x01, 1: 0x02, 2: 0x04, 3: 0x08}     # regular, stiff, guard, ill-conditioned
PLAN_FORCING_PIECEWISE, PLAN_FORCING_VARYING, PLAN_FORCING_RUNS = 0x10, 0x20, 0x80
PLAN_ROWS_ORDERED = 0x40
PLAN_ILLCOND_BLOCKS_SHIFT, PLAN_ILLCOND_BLOCKS_MAX = 12, 0x7ffff    # ABI 7: class-3 blocks counted by the plan
LITERAL_FORM_AUTO, LITERAL_FORM_ROWS, LITERAL_FORM_LANES = 0, 1, 2
STATUS_SLICE_TIMEOUT, STATUS_STALE_PLAN, STATUS_NONFINITE_FORCING = 0x1, 0x2, 0x4

_dp = ctypes.c_void_p   # device or host address, passed as an integer


class SmartEnsemble(ctypes.Structure):
    """struct SmartEnsemble of include/smart_amd.h (field for field)."""
    _fields_ = [
        ('n_catchments', ctypes.c_int64), ('n_samples', ctypes.c_int64), ('n_steps', ctypes.c_int64),
        ('n_warm', ctypes.c_int64), ('report_gap', ctypes.c_int64),
        ('report_type', ctypes.c_int32), ('math_mode', ctypes.c_int32), ('delta_sec', ctypes.c_double),
        ('area_m2', _dp), ('forcing', _dp), ('params', _dp), ('params_catchment_stride', ctypes.c_int64),
        ('extra', _dp), ('initial', _dp), ('obs', _dp), ('gw_obs', _dp),
        ('discharge', _dp), ('discharge_ld', ctypes.c_int64), ('gw', _dp), ('objfn', _dp),
        ('final_vars', _dp), ('workspace', _dp), ('workspace_bytes', ctypes.c_int64), ('stream', _dp),
        ('time_slices', ctypes.c_int32), ('plan', ctypes.c_int32),
        ('literal_form', ctypes.c_int32), ('reserved0', ctypes.c_int32),
    ]


# every symbol include/smart_amd.h declares: (restype, argtypes)
SYMBOLS = {
    'smart_n_reports': (ctypes.c_int64, [ctypes.c_int64, ctypes.c_int64, ctypes.c_int32]),
    'smart_run_ensemble_hip': (ctypes.c_int, [ctypes.POINTER(SmartEnsemble)]),
    'smart_check_ensemble': (ctypes.c_int, [ctypes.POINTER(SmartEnsemble)]),
    'smart_workspace_bytes': (ctypes.c_int64, [ctypes.POINTER(SmartEnsemble)]),
    'smart_plan_ensemble': (ctypes.c_int, [ctypes.POINTER(SmartEnsemble), ctypes.POINTER(ctypes.c_int32)]),
    'smart_launch_status': (ctypes.c_int, [ctypes.POINTER(SmartEnsemble), ctypes.POINTER(ctypes.c_int32)]),
    'smart_describe_launch': (ctypes.c_int, [ctypes.POINTER(SmartEnsemble), ctypes.c_char_p, ctypes.c_int64]),
    'smart_allsteps_hip': (ctypes.c_int, [ctypes.c_double, ctypes.c_double, ctypes.c_int64, _dp, _dp, _dp, _dp,
                                          ctypes.c_int32, ctypes.c_int64, _dp, _dp, _dp]),
    'smart_hook_counters': (ctypes.c_int, [ctypes.POINTER(ctypes.c_int64), ctypes.c_int64]),
    'smart_onestep_hip': (ctypes.c_int, [ctypes.c_int64, _dp, _dp]),
    'smart_river_step_hip': (ctypes.c_int, [ctypes.c_int64, _dp, _dp]),
    'smart_objfn_hip': (ctypes.c_int, [ctypes.c_int64, ctypes.c_int64, _dp, ctypes.c_int64, _dp, _dp,
                                       ctypes.c_double, _dp, _dp]),
    'smart_db_append_rows': (ctypes.c_int, [ctypes.c_char_p, ctypes.POINTER(ctypes.c_float), ctypes.c_int64,
                                            ctypes.c_int64, ctypes.c_int32]),
    'smart_db_parse_rows': (ctypes.c_int64, [ctypes.c_char_p, ctypes.c_int64, ctypes.c_int64,
                                             ctypes.POINTER(ctypes.c_int32), ctypes.c_int32,
                                             ctypes.POINTER(ctypes.c_float), ctypes.c_int64, ctypes.c_int32]),
    'smart_row_class': (ctypes.c_int, [_dp, ctypes.c_double, _dp, ctypes.c_double]),
    'smart_device_count': (ctypes.c_int, []),
    'smart_abi_version': (ctypes.c_int, []),
    'smart_build_info': (ctypes.c_char_p, []),
    'smart_last_error': (ctypes.c_char_p, []),
}

_lib = None


class SmartEngineError(Exception):
    """Raised when the C ABI reports an error (the reference raises plain Exception at the same places)."""

    def __init__(self, code, message):
        Exception.__init__(self, message)
        self.code = code


def _apply_lint_verdict():
    """The streaming step loops jump through byte offsets that another kernel wrote (smart_fast_arms.h: pair blocks); the
    offsets are right for a library whose code smartpy_amd.isa_lint has looked at -- smartpy_amd.build does that for
    every library it links and leaves the outcome next to it.  A library without that record, or with one written for
    another file (rebuilt elsewhere, by another hipcc), or whose blocks failed the lint, runs the THREADED CHUNKS instead
    (SMART_PAIR_BLOCKS=0: the same arithmetic, bit for bit, without computed jumps) and says so; never wrong code
    words.  A library whose hand-over or row chains failed the lint is refused.  SMART_PAIR_BLOCKS set by the caller
    stands (A/B builds: tools/variants)."""
    import warnings
    from . import isa_lint
    ok, pairs, reason = isa_lint.verdict_for(LIB_PATH)
    unrecorded = isa_lint.read_sidecar(LIB_PATH) is None or 'another build' in reason
    if ok and not pairs and unrecorded and os.path.exists(isa_lint.OBJDUMP):
        # no record, or one for another file (a library that travelled without it, or was rebuilt by something else): look
        # at the code now (~6 s) and leave the record for the next load.  ONE process of a job does it -- the others wait
        # at the lock and read what it wrote (round 5: the N ranks of a launch each disassembled the library: advisor)
        lock = None
        try:
            import fcntl
            lock = open(LIB_PATH + '.lint.lock', 'w')
            fcntl.flock(lock, fcntl.LOCK_EX)
        except (ImportError, OSError):
            lock = None         # (a read-only tree: every process looks for itself, as before)
        try:
            ok, pairs, reason = isa_lint.verdict_for(LIB_PATH)
            if ok and not pairs and (isa_lint.read_sidecar(LIB_PATH) is None or 'another build' in reason):
                report = isa_lint.check_library(LIB_PATH)
                report['stamp'] = isa_lint.library_stamp(LIB_PATH)
                try:
                    isa_lint.write_sidecar(report, LIB_PATH)
                except OSError:
                    pass
                if report['checked']:
                    if report['handover'] is False or report['rows'] is False:
                        ok, pairs, reason = False, False, '; '.join(report['problems'])
                    else:
                        ok, pairs, reason = True, bool(report['pair_blocks']), '; '.join(report['problems'])
        finally:
            if lock is not None:
                lock.close()
    if not ok:
        raise ImportError("smartpy_amd: %s failed the code lints of smartpy_amd.isa_lint (%s); rebuild it with "
                          "`python -m smartpy_amd.build --force`" % (LIB_PATH, reason))
    if 'SMART_PAIR_BLOCKS' in os.environ:
        return                  # (the caller's word stands: A/B builds, tools/variants)
    if not pairs:
        os.environ['SMART_PAIR_BLOCKS'] = '0'
        warnings.warn("smartpy_amd: %s; the step loops run their threaded chunks instead of the pair blocks "
                      "(SMART_PAIR_BLOCKS=0).  `python -m smartpy_amd.build --force` builds and checks the library."
                      % reason)
    elif isa_lint.library_stamp(LIB_PATH) != 'pairs-ok':
        # the code has been looked at (just now, or by the record's writer) but the file carries no stamp of it -- a
        # library that smartpy_amd.build did not link: the library by itself would run the threaded chunks (smart_capi.hip:
        # pair_blocks_wanted); this process has the lint's word for the pair blocks
        os.environ['SMART_PAIR_BLOCKS'] = '1'


def lib():
    """Load libsmart_amd.so once.  No fallback of any kind."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "smartpy_amd: the HIP extension %s has not been built.  Run `python -m smartpy_amd.build` "
                "(needs hipcc; cross-compiles for gfx950 without a GPU).  There is no CPU fallback." % LIB_PATH)
        # torch first: its wheel brings a HIP runtime of its own, and the one that is loaded first is the one that
        # owns the device.  Loaded the other way round (this library, then torch: __graft_entry__.build() followed by
        # smoke() in ONE process did that) the library's own runtime reports "no ROCm-capable device".
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        _apply_lint_verdict()
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)       # AttributeError here = the header and the library disagree
            fn.restype = res
            fn.argtypes = args
        if L.smart_abi_version() != ABI_VERSION:
            raise ImportError("smartpy_amd: ABI version mismatch between smartpy_amd/_lib.py and %s" % LIB_PATH)
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise SmartEngineError(rc, lib().smart_last_error().decode('utf8', 'replace') or 'error %d' % rc)
