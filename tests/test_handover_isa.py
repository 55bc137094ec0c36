"""The time-slice hand-over of the sliced kernels, checked in the code the GPU will run: the built library's gfx950
code objects are disassembled (tools/isa_report.py; hipcc cross-compiles here, no GPU needed) and every publish and
every wait must have the instruction sequence MI355X_MICROARCH.md prescribes for a plain-store payload behind a flag:

  producer   s_waitcnt vmcnt(0)  ->  buffer_wbl2 sc1  ->  s_waitcnt vmcnt(0)  ->  global_store_dword ... sc1 (the flag)
  consumer   global_load_dword ... sc1 (the poll)  ->  s_waitcnt vmcnt(0)  ->  buffer_inv sc1  ->  plain loads

hipcc (ROCm 7.2) drops the wait behind buffer_wbl2 when it can prove the wave's vmcnt scoreboard empty -- an edit far
from smart_device.h::publish_slice can trigger that; publish_slice therefore carries the wait as inline asm, and this
test is what notices the day it is gone (round-3 verdict, "What's weak" 8)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))

OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'
pytestmark = pytest.mark.skipif(not os.path.exists(OBJDUMP), reason='needs llvm-objdump of the ROCm toolchain')

# every __global__ that calls run_ensemble_merged / a sliced body (smart_fast_entry.h)
SLICED = ['smart_fast_intervals_exits', 'smart_fast_intervals', 'smart_fast_intervals_states', 'smart_fast_runs_exits',
          'smart_fast_runs', 'smart_fast_runs_states', 'smart_fast_steps', 'smart_fast_steps_states',
          'smart_fast_steps_raw', 'smart_fast_intervals_raw', 'smart_fast_steps_every']


def _all_fast_kernels():
    """the names the library itself lists (kFastKernelNames in smart_capi.hip)"""
    import re
    text = open(os.path.join(ROOT, 'smartpy_amd', 'csrc', 'smart_capi.hip')).read()
    table = re.search(r'kFastKernelNames\[kNumFastKernels\] = \{(.*?)\};', text, re.S).group(1)
    return re.findall(r'"(smart_fast_\w+)"', table)


def _kernel(name):
    import isa_report
    lib = os.path.join(ROOT, 'smartpy_amd', 'csrc', 'libsmart_amd.so')
    start, symbol, body = isa_report.disassemble(lib, name)
    return isa_report.parse(start, body)


def _is_wait_vm0(x):
    return x['op'] == 's_waitcnt' and 'vmcnt(0)' in x['args']


def _is_payload_access(x):
    """a vector-memory access that is neither a flag access (sc1) nor an atomic"""
    return x['cls'] == 'vmem' and x['op'].startswith(('global_load', 'global_store', 'flat_load', 'flat_store',
                                                      'buffer_load', 'buffer_store')) and 'sc1' not in x['args']


@pytest.mark.parametrize('kernel', SLICED)
def test_publish_and_wait_sequences_of_the_sliced_kernels(kernel):
    insts = _kernel(kernel)
    releases = [i for i, x in enumerate(insts) if x['op'] == 'buffer_wbl2']
    polls = [i for i, x in enumerate(insts) if x['op'] == 'global_load_dword' and 'sc1' in x['args']]
    # a sliced body publishes in two places (a slice that ran; a slice that gave up and poisons its chain) and waits
    # in one; the compiler may duplicate either, it may not lose one
    assert len(releases) >= 2 and len(polls) >= 1, (len(releases), len(polls))
    for i in releases:
        assert 'sc1' in insts[i]['args']                                    # agent scope, not workgroup
        # straight-line from the write-back to the flag store: a wait for it, and nothing that signals before the wait
        j = i + 1
        waited = False
        while j < len(insts) and not (insts[j]['op'].startswith(('global_store', 'global_atomic')) and
                                      ('sc1' in insts[j]['args'] or insts[j]['op'].startswith('global_atomic'))):
            waited = waited or _is_wait_vm0(insts[j])
            assert waited or insts[j]['cls'] not in ('vmem', 'branch'), \
                '%s: %s %s between buffer_wbl2 and its wait' % (kernel, insts[j]['op'], insts[j]['args'])
            j += 1
        assert j < len(insts) and waited, '%s: no s_waitcnt vmcnt(0) between buffer_wbl2 and the flag store' % kernel
        assert insts[j]['op'] == 'global_store_dword' and 'sc1' in insts[j]['args']     # the relaxed agent flag store
        # ... and ahead of the write-back the wave's own payload stores have been waited for
        k = i - 1
        while k >= 0 and insts[k]['cls'] not in ('vmem', 'branch'):
            if _is_wait_vm0(insts[k]):
                break
            k -= 1
        assert k >= 0 and _is_wait_vm0(insts[k]), '%s: payload stores not drained ahead of buffer_wbl2' % kernel
    for i in polls:
        # behind the poll (in address order: the exit of the poll loop lies behind it) the L1 invalidate comes before
        # the first plain load of the hand-over; its own wait stands directly in front of it
        j = i + 1
        while j < len(insts) and insts[j]['op'] != 'buffer_inv':
            assert not _is_payload_access(insts[j]), \
                '%s: %s %s between the poll and buffer_inv' % (kernel, insts[j]['op'], insts[j]['args'])
            j += 1
        assert j < len(insts) and 'sc1' in insts[j]['args'], '%s: no buffer_inv sc1 behind the poll' % kernel
        assert _is_wait_vm0(insts[j - 1])


def test_no_sliced_kernel_is_missing_from_the_list_above():
    """a kernel that publishes (buffer_wbl2 in its code) is a sliced kernel and must be linted"""
    names = _all_fast_kernels()
    assert len(names) >= 12 and set(SLICED) <= set(names)
    for name in names:
        publishes = any(x['op'] == 'buffer_wbl2' for x in _kernel(name))
        assert publishes == (name in SLICED), name
