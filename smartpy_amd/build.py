"""Build libsmart_amd.so (HIP, gfx950) in-tree with hipcc.  `python -m smartpy_amd.build [--force]`.

hipcc cross-compiles for gfx950 without a GPU.  The shared object stays next to the sources
(smartpy_amd/csrc/libsmart_amd.so) so that it travels with the tree; it is git-ignored.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(CSRC, 'libsmart_amd.so')
ARCH = 'gfx950'

# translation unit -> extra flags.  The literal kernels must round every operation separately.
UNITS = {
    'smart_literal.hip': ['-ffp-contract=off'],
    # the literal model inside the fast kernel opts out of contraction per function (pragma); no NaN ever enters the
    # arithmetic (missing observations are tested on their bit pattern), which spares the sNaN-quieting
    # `v_max_f64 x, x, x` hipcc otherwise puts in front of fmin / fmax (-1.2 % kernel time, A/B measured)
    'smart_fast_intervals.hip': ['-ffp-contract=fast-honor-pragmas', '-fno-honor-nans'],
    'smart_fast_runs.hip': ['-ffp-contract=fast-honor-pragmas', '-fno-honor-nans'],
    'smart_fast_steps.hip': ['-ffp-contract=fast-honor-pragmas', '-fno-honor-nans'],
    'smart_fast_reports.hip': ['-ffp-contract=fast-honor-pragmas', '-fno-honor-nans'],
    # ... except where the literal model lives inside the fast mode (the ill-conditioned rows and rows with NaN or
    # infinite parameters): NaNs honoured -- what a NaN does in the reference's compares is part of what it reproduces
    'smart_fast_guarded.hip': ['-ffp-contract=fast-honor-pragmas'],
    'smart_capi.hip': [],
    'smart_hostio.cpp': ['-pthread'],      # host only: the sampling-database writer
}
COMMON = ['-O3', '-fPIC', '-std=c++17', '--offload-arch=' + ARCH, '-fno-gpu-rdc', '-Wall']
DEPS = ['smart_device.h', 'smart_literal_model.h', 'smart_literal_lanes.h', 'smart_fast_model.h', 'smart_fast_arms.h', 'smart_fast_entry.h', os.path.join('..', '..', 'include', 'smart_amd.h')]


def hipcc():
    exe = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(exe):
        raise RuntimeError('hipcc not found: the HIP extension cannot be built')
    return exe


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build(force=False, verbose=False, extra_flags=(), lib_path=LIB):
    cc = hipcc()
    deps = [os.path.join(CSRC, d) for d in DEPS] + [os.path.abspath(__file__)]
    objs, jobs = [], []
    suffix = '' if lib_path == LIB else '.' + os.path.basename(lib_path)
    for unit, flags in UNITS.items():
        src = os.path.join(CSRC, unit)
        obj = os.path.join(CSRC, os.path.splitext(unit)[0] + suffix + '.o')
        if force or _stale(obj, [src] + deps):
            jobs.append([cc] + COMMON + flags + list(extra_flags) + ['-c', src, '-o', obj])
        objs.append(obj)
    if jobs:        # the translation units are independent: compile them side by side
        if verbose:
            for cmd in jobs:
                print(' '.join(cmd))
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as pool:
            list(pool.map(subprocess.check_call, jobs))
    if force or _stale(lib_path, objs):
        # linked under a name of this process's own, then renamed onto the library: a process that loads it meanwhile sees
        # the old file or the new one, never half of one
        tmp = '%s.%d.tmp' % (lib_path, os.getpid())
        cmd = [cc, '-shared', '-fPIC', '-pthread', '--offload-arch=' + ARCH, '-o', tmp] + objs
        if verbose:
            print(' '.join(cmd))
        try:
            subprocess.check_call(cmd)
            from . import isa_lint
            report = lint(tmp, verbose)
            # the verdict on the pair blocks goes INTO the file (a caller of the C ABI reads no record next to it), then
            # the record is written for the file as it now is
            isa_lint.stamp_library(tmp, bool(report['pair_blocks']))
            report['sha256'] = isa_lint.sha256_of(tmp)
            report['stamp'] = isa_lint.library_stamp(tmp)
            os.replace(tmp, lib_path)
            isa_lint.write_sidecar(report, lib_path)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
    elif lint_missing(lib_path):
        # a library without a record of its own (copied here, or its record lost): looked at now, stamped in a copy that
        # then takes its place (a process that has the old file loaded keeps it)
        from . import isa_lint
        report = lint(lib_path, verbose)
        tmp = '%s.%d.tmp' % (lib_path, os.getpid())
        try:
            shutil.copy(lib_path, tmp)
            isa_lint.stamp_library(tmp, bool(report['pair_blocks']))
            report['sha256'] = isa_lint.sha256_of(tmp)
            report['stamp'] = isa_lint.library_stamp(tmp)
            os.replace(tmp, lib_path)
            isa_lint.write_sidecar(report, lib_path)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
    return lib_path


class BuildLintError(RuntimeError):
    pass


def lint_missing(lib_path):
    from . import isa_lint
    rep = isa_lint.read_sidecar(lib_path)
    return rep is None or rep.get('sha256') != isa_lint.sha256_of(lib_path)


def lint(path, verbose=False):
    """smartpy_amd.isa_lint on a freshly linked library, BEFORE it is put in place: the pair blocks where the code words
    point, the hand-over sequences, the DPP distances of the row form.  A hand-over or a row chain that fails refuses
    the library (BuildLintError: those have no other form to fall back to); pair blocks that fail are recorded, and the
    library will run its threaded chunks (smartpy_amd._lib)."""
    from . import isa_lint
    report = isa_lint.check_library(path)
    report['library'] = os.path.basename(LIB)
    if verbose or report['problems']:
        print('isa_lint: %s' % ('; '.join(report['problems']) if report['problems'] else
                                'pair blocks, hand-over and row chains of the linked library are in order'))
    if report['handover'] is False or report['rows'] is False:
        raise BuildLintError('smartpy_amd.build: the linked library fails the code lints and is not installed: '
                             + '; '.join(report['problems']))
    return report


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
