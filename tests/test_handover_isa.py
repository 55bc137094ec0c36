"""The time-slice hand-over of the sliced kernels, checked in the code the GPU will run: the built library's gfx950
code objects are disassembled (smartpy_amd/isa_lint.py; hipcc cross-compiles here, no GPU needed) and every publish and
every wait must have the instruction sequence MI355X_MICROARCH.md prescribes for a plain-store payload behind a flag:

  producer   s_waitcnt vmcnt(0)  ->  buffer_wbl2 sc1  ->  s_waitcnt vmcnt(0)  ->  global_store_dword ... sc1 (the flag)
  consumer   global_load_dword ... sc1 (the poll)  ->  s_waitcnt vmcnt(0)  ->  buffer_inv sc1  ->  plain loads

hipcc (ROCm 7.2) drops the wait behind buffer_wbl2 when it can prove the wave's vmcnt scoreboard empty -- an edit far
from smart_device.h::publish_slice can trigger that; publish_slice therefore carries the wait as inline asm, and this
lint is what notices the day it is gone (round-3 verdict, "What's weak" 8).  Since round 5 smartpy_amd.build runs it on
every library it links and refuses one that fails."""
import os

import pytest

from smartpy_amd import isa_lint

pytestmark = pytest.mark.skipif(not os.path.exists(isa_lint.OBJDUMP), reason='needs llvm-objdump of the ROCm toolchain')


@pytest.fixture(scope='module')
def dis():
    return isa_lint.Disassembly(isa_lint.LIB)


@pytest.mark.parametrize('kernel', isa_lint.SLICED)
def test_publish_and_wait_sequences_of_the_sliced_kernels(dis, kernel):
    isa_lint.lint_handover(dis, kernel)


def test_no_sliced_kernel_is_missing_from_the_list(dis):
    """a kernel that publishes (buffer_wbl2 in its code) is a sliced kernel and must be linted"""
    isa_lint.lint_handover(dis)


def test_the_lint_notices_a_missing_wait(dis):
    """the same check on a doctored instruction stream: the wait behind buffer_wbl2 taken out"""
    insts = [dict(x) for x in dis.kernel('smart_fast_steps')]
    i = next(k for k, x in enumerate(insts) if x['op'] == 'buffer_wbl2')
    j = i + 1
    while not (insts[j]['op'] == 'global_store_dword' and 'sc1' in insts[j]['args']):     # up to the flag store
        if insts[j]['op'] == 's_waitcnt' and 'vmcnt(0)' in insts[j]['args']:
            insts[j] = dict(insts[j], op='s_nop', args='0', cls='other')
        j += 1

    class Doctored(isa_lint.Disassembly):
        def __init__(self):
            pass

        def kernel(self, name):
            return insts

    with pytest.raises(isa_lint.LintError, match='buffer_wbl2'):
        isa_lint.lint_handover(Doctored(), 'smart_fast_steps')
