"""The DPP chains of the row form of the literal step (smartpy_amd/csrc/smart_literal_lanes.h), checked in the code the
GPU will run.  The chains are inline asm, which hipcc's hazard recogniser does not look into: the two wait states
gfx950 wants between a vector instruction that writes a register and a DPP instruction that reads it are kept by the
layout of the asm blocks themselves (no s_nop: each would cost a lone wavefront a real instruction's four cycles).
smartpy_amd.isa_lint.lint_rows walks every DPP instruction of the two kernels that hold the row form and looks at
what can execute in the two wait states ahead of it, along fall-through and branches."""
import os

import pytest

from smartpy_amd import isa_lint

pytestmark = pytest.mark.skipif(not os.path.exists(isa_lint.OBJDUMP), reason='needs llvm-objdump of the ROCm toolchain')


@pytest.fixture(scope='module')
def dis():
    return isa_lint.Disassembly(isa_lint.LIB)


@pytest.mark.parametrize('kernel', isa_lint.ROWS)
def test_no_dpp_read_within_two_wait_states_of_its_register_being_written(dis, kernel):
    isa_lint.lint_rows(dis, kernel)


def test_the_hot_step_of_the_row_form_carries_no_s_nop_and_no_scratch(dis):
    """the blocks of the quick step (those that hold a run of a dozen DPP instructions): no s_nop of the asm's own, no
    scratch access, no call -- the guarded form's call and its buffer live in other blocks"""
    for kernel in isa_lint.ROWS:
        insts = dis.kernel(kernel)
        cuts = [i for i, x in enumerate(insts) if x['cls'] == 'branch' or x['op'] in ('s_endpgm', 's_setpc_b64', 's_swappc_b64')]
        hot = 0
        for lo, hi in zip([0] + [c + 1 for c in cuts], cuts + [len(insts)]):
            block = insts[lo:hi]
            if sum(x['op'] == 'v_fmac_f64_dpp' for x in block) < 12:        # (the chains; not get_vars' seventeen broadcasts)
                continue
            hot += 1
            assert not any(x['op'].startswith('scratch_') for x in block), kernel
            for k, x in enumerate(block):       # an s_nop directly in front of a DPP instruction would be one of the asm's own
                if x['op'] == 's_nop' and k + 1 < len(block):
                    assert 'row_newbcast' not in block[k + 1]['args'], (kernel, hex(x['addr']))
        assert hot >= 8, (kernel, hot)


def test_the_lint_notices_a_dpp_read_right_behind_its_producer():
    """a doctored stream: v_mul_f64 writes v[4:5], one instruction later a v_fmac_f64_dpp reads them through DPP"""
    def inst(addr, op, args):
        return {'addr': addr, 'op': op, 'args': args, 'cls': isa_lint.classify(op), 'target': None, 'size': 8}
    filler = [inst(8 * k, 'v_fmac_f64_dpp', 'v[0:1], v[2:3], v[8:9] row_newbcast:%d row_mask:0xf bank_mask:0xf' % (k % 6))
              for k in range(100)]
    tail = [inst(800, 'v_mul_f64', 'v[4:5], v[6:7], v[6:7]'),
            inst(808, 'v_add_f64', 'v[10:11], v[6:7], v[6:7]'),
            inst(816, 'v_fmac_f64_dpp', 'v[0:1], v[4:5], v[8:9] row_newbcast:0 row_mask:0xf bank_mask:0xf')]

    class Fake(isa_lint.Disassembly):
        def __init__(self, insts):
            self.insts = insts

        def kernel(self, name):
            return self.insts

    with pytest.raises(isa_lint.LintError, match='1 wait state'):
        isa_lint.lint_rows(Fake(filler + tail), 'smart_fast_illcond')
    ok = filler + tail[:2] + [inst(816, 'v_add_f64', 'v[12:13], v[6:7], v[6:7]'), dict(tail[2], addr=824)]
    isa_lint.lint_rows(Fake(ok), 'smart_fast_illcond')
    # s_nop 1 counts for two wait states; a branch into the DPP instruction is followed back to where it came from
    nop = filler + [tail[0], inst(808, 's_nop', '1'), tail[2]]
    isa_lint.lint_rows(Fake(nop), 'smart_fast_illcond')
    jump = filler + [tail[0], dict(inst(808, 's_branch', '12'), target=832), inst(816, 'v_add_f64', 'v[12:13], v[6:7], v[6:7]'),
                     inst(824, 'v_add_f64', 'v[14:15], v[6:7], v[6:7]'), dict(tail[2], addr=832)]
    with pytest.raises(isa_lint.LintError, match='wait state'):
        isa_lint.lint_rows(Fake(jump), 'smart_fast_illcond')
