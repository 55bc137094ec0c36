"""The pair blocks of the streaming step loop (smart_fast_arms.h: SMART_A_PAIRS_STRETCH), checked in the code the GPU
will run: smart_forcing_scan's code words are byte offsets into that code, computed from a stride and a block order
that the asm has to honour -- a block that outgrew its room or changed its place would send a jump into the middle of
another.  The built library is disassembled (hipcc cross-compiles here, no GPU needed) and every block looked at."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))

OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'
pytestmark = pytest.mark.skipif(not os.path.exists(OBJDUMP), reason='needs llvm-objdump of the ROCm toolchain')

KINDS = 'CDR'          # smart_device.h: step_kind -- 0 calm, 1 dry, 2 rain
FIRST = {'C': 'v_cmp_lt_f64', 'D': 'v_mul_f64', 'R': 'v_mov_b64'}      # how the three arms begin


def _stride(name='SMART_P_STRIDE'):
    text = open(os.path.join(ROOT, 'smartpy_amd', 'csrc', 'smart_device.h')).read()
    return int(re.search(r'#define %s (\d+)' % name, text).group(1))


def _kernel(name):
    import isa_report
    lib = os.path.join(ROOT, 'smartpy_amd', 'csrc', 'libsmart_amd.so')
    start, symbol, body = isa_report.disassemble(lib, name)
    insts = isa_report.parse(start, body)
    for a, b in zip(insts, insts[1:]):
        a['size'] = b['addr'] - a['addr']
    insts[-1]['size'] = 4
    return insts


@pytest.mark.parametrize('kernel', ['smart_fast_steps', 'smart_fast_steps_raw', 'smart_fast_steps_states'])
def test_every_pair_block_lies_where_the_code_words_point(kernel):
    split = kernel.endswith('_states')      # the models with the final state vector: larger blocks, no stream of records
    stride = _stride('SMART_PS_STRIDE' if split else 'SMART_P_STRIDE')
    assert stride % 64 == 0
    insts = _kernel(kernel)
    at = {x['addr']: i for i, x in enumerate(insts)}
    entries = [i for i, x in enumerate(insts) if x['op'] == 's_getpc_b64' and 's[78:79]' in x['args']]
    # the stream of records (SMART_A_GAP_STREAM) loads ONE code word per pair, the pair blocks two per chunk
    pairs = [i for i in entries if not any(x['op'] == 's_load_dword' for x in insts[i:i + 12])]
    assert len(pairs) == 2, 'two instances of the stretch asm per kernel: even / any number of chunks per interval'
    for i in pairs:
        _check_instance(insts, at, i, stride)
    streams = [i for i in entries if i not in pairs]
    assert len(streams) == (0 if split else 1)
    if streams:
        _check_gap_stream(insts, at, streams[0], _stride('SMART_E_STRIDE'))


def _check_gap_stream(insts, at, i, stride):
    """54 blocks: 2 buffers x 3 variants (no report in the pair / behind its first arm / behind its second) x 9 patterns;
    every main path ends with the computed jump inside the block's room, requests one pair, and holds the report's
    row-pointer move exactly where its variant says"""
    base = insts[i]['addr'] + 4 + int(insts[i + 1]['args'].split(',')[-1], 0)
    assert base % 64 == 0
    for n in range(54):
        variant, pattern = (n % 27) // 9, n % 9
        names = KINDS[pattern // 3] + KINDS[pattern % 3]
        b = base + n * stride
        assert b in at, 'block %d does not start on an instruction' % n
        k = at[b]
        if names[0] == 'R':
            assert insts[k]['op'] == 's_nop' and insts[k]['size'] == 4
            k += 1
        assert insts[k]['op'].startswith(FIRST[names[0]]), (n, names, insts[k]['op'])
        loads = reports = 0
        while insts[k]['op'] != 's_setpc_b64':
            assert insts[k]['addr'] < b + stride, 'block %d (%s) outgrew its %d bytes' % (n, names, stride)
            loads += insts[k]['op'].startswith('s_load_dwordx16')
            reports += insts[k]['op'] == 'v_lshl_add_u64'
            k += 1
        assert loads == 1 and reports == (variant != 0), (n, names, variant, loads, reports)


def _check_instance(insts, at, i, stride):
    assert insts[i + 1]['op'] == 's_add_u32' and insts[i + 1]['args'].startswith('s78, s78,')
    base = insts[i]['addr'] + 4 + int(insts[i + 1]['args'].split(',')[-1], 0)
    assert base % 64 == 0
    # the entry (the second form has two: a stretch may start in either buffer) ends with the jump to the first block,
    # nothing falls into the blocks
    j = i
    while insts[j]['addr'] < base:
        last = insts[j]
        j += 1
    while last['op'] == 's_nop':
        j -= 1
        last = insts[j - 1]
    assert last['op'] == 's_setpc_b64'

    def block(n, names, tail_loads):
        b = base + n * stride
        assert b in at, 'block %d does not start on an instruction' % n
        k = at[b]
        if names[0] == 'R':         # entered 4 bytes in (pair_code adds 4): an s_nop on the boundary
            assert insts[k]['op'] == 's_nop' and insts[k]['size'] == 4
            k += 1
        assert insts[k]['op'].startswith(FIRST[names[0]]), (n, names, insts[k]['op'])
        # the main path: up to the computed jump
        loads = 0
        while insts[k]['op'] != 's_setpc_b64':
            assert insts[k]['addr'] < b + stride, 'block %d (%s) outgrew its %d bytes' % (n, names, stride)
            loads += insts[k]['op'].startswith('s_load_dwordx16')
            k += 1
        assert insts[k]['args'].strip() == 's[76:77]'
        assert loads == tail_loads, (n, names, loads)
        # what follows the jump (out-of-line cascades) stays inside the block's room and ends with a branch back
        end = insts[k]['addr'] + 4
        m = k + 1
        while m < len(insts) and insts[m]['addr'] < b + stride:
            if insts[m]['op'].startswith(('v_', 's_branch', 's_cbranch')):
                end = insts[m]['addr'] + insts[m]['size']
            m += 1
        assert end <= b + stride or n == 39      # (behind the last block: the report blocks, not bound to its room)
        return k

    n = 0
    for pos in range(4):                    # buffer 0: first pair, second pair; buffer 1: first, second
        for k0 in KINDS:
            for k1 in KINDS:
                block(n, k0 + k1, tail_loads=pos % 2)       # the second pair's tail requests the chunk after next
                n += 1
    for buf in range(2):                    # whole chunks: four calm, four dry steps
        for k0 in 'CD':
            block(n, k0 * 4, tail_loads=1)
            n += 1
    assert n == 40


def test_a_dry_pair_is_eighteen_instructions_on_the_boundary():
    """the shortest block, DD of the first position: 2 x 9 vector instructions, all 64-bit encodings on 8-byte addresses,
    then the two instructions of the jump"""
    stride = _stride()
    insts = _kernel('smart_fast_steps')
    at = {x['addr']: i for i, x in enumerate(insts)}
    i = [k for k, x in enumerate(insts) if x['op'] == 's_getpc_b64' and 's[78:79]' in x['args'] and
         not any(y['op'] == 's_load_dword' for y in insts[k:k + 12])][0]
    base = insts[i]['addr'] + 4 + int(insts[i + 1]['args'].split(',')[-1], 0)
    k = at[base + 4 * stride]
    body = insts[k:k + 20]
    assert [x['size'] for x in body[:18]] == [8] * 18 and all(x['addr'] % 8 == 0 for x in body[:18])
    assert all(x['op'].startswith(('v_mul_f64', 'v_fma_f64', 'v_add_f64')) for x in body[:18])
    assert [x['op'] for x in body[18:20]] == ['s_add_u32', 's_setpc_b64']


def test_every_block_of_the_every_step_stream_lies_where_its_code_word_points():
    """SMART_A_EVERY_STREAM (a report every step): four instances in smart_fast_steps_every (matrix stored or not,
    observations or not), each 2 x 9 blocks SMART_E_STRIDE bytes apart; every block's main path -- arm, report, arm,
    report, loop control -- ends with the computed jump inside the block's room and requests exactly one pair of steps"""
    stride = _stride('SMART_E_STRIDE')
    assert stride % 64 == 0
    insts = _kernel('smart_fast_steps_every')
    at = {x['addr']: i for i, x in enumerate(insts)}
    entries = [i for i, x in enumerate(insts) if x['op'] == 's_getpc_b64' and 's[78:79]' in x['args']]
    assert len(entries) == 4
    stores = []
    for i in entries:
        base = insts[i]['addr'] + 4 + int(insts[i + 1]['args'].split(',')[-1], 0)
        assert base % 64 == 0
        n_store = 0
        for n in range(18):
            names = KINDS[(n % 9) // 3] + KINDS[n % 3]
            b = base + n * stride
            assert b in at, 'block %d does not start on an instruction' % n
            k = at[b]
            if names[0] == 'R':
                assert insts[k]['op'] == 's_nop' and insts[k]['size'] == 4
                k += 1
            assert insts[k]['op'].startswith(FIRST[names[0]]), (n, names, insts[k]['op'])
            loads = sums = 0
            while insts[k]['op'] != 's_setpc_b64':
                assert insts[k]['addr'] < b + stride, 'block %d (%s) outgrew its %d bytes' % (n, names, stride)
                loads += insts[k]['op'].startswith('s_load_dwordx16')
                n_store += insts[k]['op'] == 'global_store_dwordx2'
                k += 1
            assert loads == 1, (n, names)
        stores.append(n_store)
    assert sorted(stores) == [0, 0, 36, 36]         # two of the four instances store, one value per step
