"""Build libsmart_amd.so (HIP, gfx950) in-tree with hipcc.  `python -m smartpy_amd.build [--force]`.

hipcc cross-compiles for gfx950 without a GPU.  The shared object stays next to the sources
(smartpy_amd/csrc/libsmart_amd.so) so that it travels with the tree; it is git-ignored.
"""
import os
import shutil
import subprocess
import sys

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
    'smart_fast.hip': ['-ffp-contract=fast-honor-pragmas', '-fno-honor-nans'],
    'smart_capi.hip': [],
    'smart_hostio.cpp': ['-pthread'],      # host only: the sampling-database writer
}
COMMON = ['-O3', '-fPIC', '-std=c++17', '--offload-arch=' + ARCH, '-fno-gpu-rdc', '-Wall']
DEPS = ['smart_device.h', 'smart_literal_model.h', os.path.join('..', '..', 'include', 'smart_amd.h')]


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
    objs = []
    suffix = '' if lib_path == LIB else '.' + os.path.basename(lib_path)
    for unit, flags in UNITS.items():
        src = os.path.join(CSRC, unit)
        obj = os.path.join(CSRC, os.path.splitext(unit)[0] + suffix + '.o')
        if force or _stale(obj, [src] + deps):
            cmd = [cc] + COMMON + flags + list(extra_flags) + ['-c', src, '-o', obj]
            if verbose:
                print(' '.join(cmd))
            subprocess.check_call(cmd)
        objs.append(obj)
    if force or _stale(lib_path, objs):
        cmd = [cc, '-shared', '-fPIC', '-pthread', '--offload-arch=' + ARCH, '-o', lib_path] + objs
        if verbose:
            print(' '.join(cmd))
        subprocess.check_call(cmd)
    return lib_path


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
