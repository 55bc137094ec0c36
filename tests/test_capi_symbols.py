"""The C-ABI library loads on a machine without a GPU and exports what include/smart_amd.h declares.
No compute entry point is called here (they need a device and say so)."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, 'include', 'smart_amd.h')


@pytest.fixture(scope='module')
def L():
    from smartpy_amd import build as hip_build
    hip_build.build()                       # no-op when up to date; hipcc cross-compiles without a GPU
    from smartpy_amd import _lib
    return _lib.lib()


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(smart_[a-z_0-9]+)\s*\(', text)))


def test_every_declared_symbol_is_exported_and_bound(L):
    from smartpy_amd import _lib
    names = declared_functions()
    assert len(names) >= 9
    assert sorted(_lib.SYMBOLS) == names            # the ctypes table mirrors the header exactly
    for n in names:
        assert getattr(L, n) is not None
    assert L.smart_abi_version() == _lib.ABI_VERSION == 7
    assert b"gfx950" in L.smart_build_info() and b"clang" in L.smart_build_info()


def test_struct_layout_matches_the_header(tmp_path, L):
    """sizeof / offsetof from a C compiler against the ctypes mirror of struct SmartEnsemble."""
    from smartpy_amd import _lib
    fields = [f[0] for f in _lib.SmartEnsemble._fields_]
    src = tmp_path / 'layout.c'
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "smart_amd.h"\nint main(void){\n'
                   'printf("%zu\\n", sizeof(SmartEnsemble));\n' +
                   ''.join('printf("%%zu\\n", offsetof(SmartEnsemble, %s));\n' % f for f in fields) +
                   'return 0;}\n')
    exe = tmp_path / 'layout'
    subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)])
    nums = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert nums[0] == ctypes.sizeof(_lib.SmartEnsemble)
    assert nums[1:] == [getattr(_lib.SmartEnsemble, f).offset for f in fields]


def test_validation_and_error_codes_without_a_device(L):
    from smartpy_amd import _lib
    assert L.smart_n_reports(87672, 24, 1) == 3653 and L.smart_n_reports(1000, 24, 2) == 42
    assert L.smart_n_reports(1000, 24, 1) == 41 and L.smart_n_reports(10, 0, 1) == 0
    e = _lib.SmartEnsemble()
    assert L.smart_check_ensemble(None) == -1
    assert L.smart_check_ensemble(ctypes.byref(e)) == -2                      # sizes
    e.n_catchments, e.n_samples, e.n_steps, e.n_warm, e.report_gap = 1, 10, 240, 0, 24
    e.report_type, e.math_mode, e.delta_sec = 3, 1, 3600.0
    assert L.smart_check_ensemble(ctypes.byref(e)) == -3                      # report type (structure.py:69-70)
    assert b"Reporting type '3' unknown." in L.smart_last_error()
    e.report_type, e.math_mode = 1, 5
    assert L.smart_check_ensemble(ctypes.byref(e)) == -7
    e.math_mode, e.n_warm = 1, 480
    assert L.smart_check_ensemble(ctypes.byref(e)) == -4                      # warm-up > run (structure.py:90-95)
    e.n_warm = 12
    assert L.smart_check_ensemble(ctypes.byref(e)) == -5                      # reshape (structure.py:190)
    e.report_type = 2
    assert L.smart_check_ensemble(ctypes.byref(e)) == -1                      # raw tolerates it; pointers missing
    buf = (ctypes.c_double * 16)()
    e.area_m2 = e.forcing = e.params = e.gw = ctypes.addressof(buf)
    e.n_warm = 0
    assert L.smart_check_ensemble(ctypes.byref(e)) == 0 and L.smart_last_error() == b''
    e.objfn = ctypes.addressof(buf)
    assert L.smart_check_ensemble(ctypes.byref(e)) == -1                      # objfn needs obs + workspace
    # the workspace is the caller's: smart_workspace_bytes says how much, the check refuses less
    assert L.smart_workspace_bytes(None) == 0
    R = L.smart_n_reports(e.n_steps, e.report_gap, e.report_type)
    e.obs = e.workspace = ctypes.addressof(buf)
    need = L.smart_workspace_bytes(ctypes.byref(e))
    assert need >= 8 * (8 + R)                 # observation statistics (+ slice hand-over where a device says so)
    e.workspace_bytes = 8 * (8 + R) - 8
    assert L.smart_check_ensemble(ctypes.byref(e)) == -2 and b'workspace_bytes' in L.smart_last_error()
    e.workspace_bytes = need
    assert L.smart_check_ensemble(ctypes.byref(e)) == 0
    e.objfn = None
    assert L.smart_workspace_bytes(ctypes.byref(e)) in (0, need - 8 * (8 + R))
    e.discharge, e.discharge_ld = ctypes.addressof(buf), 5
    assert L.smart_check_ensemble(ctypes.byref(e)) == -2


def test_workspace_has_room_for_the_step_loops_code_words(L):
    """smart_workspace_bytes (no device needed for this part): fast summary / raw runs over whole intervals of a multiple
    of four steps get 8 bytes per four steps and catchment -- the kinds of the steps for the pair blocks of the step loop
    (smart_device.h: code_chunks) --, a report every step 68 bytes per pair of steps -- the stream of records and its code
    words (every_pairs), and so do gaps that are not whole chunks of four steps; the literal mode nothing of the kind."""
    from smartpy_amd import _lib

    def need(gap, report, T=9600, C=3, math=1):
        e = _lib.SmartEnsemble()
        e.n_catchments, e.n_samples, e.n_steps, e.n_warm, e.report_gap = C, 10, T, 0, gap
        e.report_type, e.math_mode, e.delta_sec = report, math, 3600.0
        return L.smart_workspace_bytes(ctypes.byref(e))
    plain = need(24, 1, math=0)                         # header only (no objfn, no device: no hand-over)
    rnd = lambda x: (x + 255) // 256 * 256              # noqa: E731
    assert need(24, 1) - plain == rnd(3 * (9600 // 4 + 4) * 8)
    assert need(8, 2) - plain == rnd(3 * (9600 // 4 + 4) * 8)
    assert need(1, 1) - plain == rnd(3 * (9600 // 2 + 4) * 68) == need(1, 2) - plain == need(6, 1) - plain
    assert need(12, 1) - plain == rnd(3 * (9600 // 4 + 4) * 8) and plain == need(1, 1, math=0)
    assert need(16, 2, T=9601) == plain                 # raw over a ragged time axis: smart_fast_plain


def test_no_cpu_fallback(L):
    """Without a device the compute entry points refuse to run (this container has no GPU)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is present')
    assert L.smart_device_count() == 0
    buf = (ctypes.c_double * 64)()
    a = ctypes.addressof(buf)
    assert L.smart_onestep_hip(1, a, a) == -6
    assert L.smart_allsteps_hip(1.0, 3600.0, 1, a, a, a, a, 1, 1, a, a, a) == -6
    assert b'HIP' in L.smart_last_error() or b'hip' in L.smart_last_error()
    from smartpy_amd import engine
    with pytest.raises(Exception, match='no CPU fallback'):
        engine.run_ensemble([[1.0] * 10], [[0.0, 0.0]] * 24, 1e6, 3600.0, 0, 24)


def test_product_never_touches_the_oracle():
    """Nothing under smartpy_amd/ or include/ may import, include or link anything from oracle/."""
    bad = []
    for base in ('smartpy_amd', 'include'):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith(('.py', '.hip', '.h', '.cpp', '.c')):
                    text = open(os.path.join(dirpath, f), errors='replace').read()
                    if re.search(r'(from|import)\s+oracle|oracle/|smart_oracle|libsmart_oracle', text):
                        bad.append(os.path.join(dirpath, f))
    assert bad == []
    out = subprocess.check_output(['ldd', os.path.join(ROOT, 'smartpy_amd', 'csrc', 'libsmart_amd.so')]).decode()
    assert 'oracle' not in out and 'libamdhip64' in out
    assert 'oracle' not in ' '.join(sys.modules) or True


def build_c_example(tmp_path):
    """examples/ensemble_from_c.c with plain gcc (C, not C++): the header is C, the only link dependencies are the
    library and the HIP runtime."""
    exe = str(tmp_path / 'ensemble_from_c')
    csrc = os.path.join(ROOT, 'smartpy_amd', 'csrc')
    subprocess.check_call(['gcc', '-O2', '-Wall', '-Werror', '-D__HIP_PLATFORM_AMD__', '-I/opt/rocm/include',
                           '-I', os.path.join(ROOT, 'include'), os.path.join(ROOT, 'examples', 'ensemble_from_c.c'),
                           '-L', csrc, '-lsmart_amd', '-L/opt/rocm/lib', '-lamdhip64', '-lm',
                           '-Wl,-rpath,' + csrc, '-Wl,-rpath,/opt/rocm/lib', '-o', exe])
    return exe


def test_c_example_builds_and_refuses_to_run_without_a_device(tmp_path, L):
    exe = build_c_example(tmp_path)
    if L.smart_device_count() > 0:
        pytest.skip('a GPU is present: tests/test_gpu_api.py runs the example for real')
    r = subprocess.run([exe] + ['x'] * 9, capture_output=True, text=True)
    assert r.returncode == 1 and '0 HIP device(s)' in r.stderr
