#!/usr/bin/env python3
"""Disassemble a kernel of the BUILT library and list its loops, instruction by instruction, with a histogram per basic
block: what the roofline figures of DESIGN.md / bench.py rest on, checkable from the tree.

    python tools/isa_report.py smart_fast_steps            profiles/r03_isa_steps
    python tools/isa_report.py smart_fast_intervals        profiles/r03_isa_intervals
    python tools/isa_report.py smart_fast_intervals_exits  profiles/r03_isa_intervals_exits

Writes <out>.s (the kernel's disassembly from llvm-objdump of the gfx950 code object inside libsmart_amd.so: every
basic block that belongs to a loop is headed by a `;; ---- block Bn [loops ...]` line with its instruction classes) and
<out>.json (blocks, loops, class counts: what tools/isa_model.py weighs with the workload's path frequencies).

Instruction classes
    fp64      v_fma_f64 v_fmac_f64 v_add_f64 v_mul_f64 v_min_f64 v_max_f64 v_ldexp_f64 (the model's arithmetic)
    vcmp      v_cmp*                 vmov   v_mov* v_cndmask* v_accvgpr*      lane  v_readlane v_writelane v_readfirstlane
    valu      any other vector ALU   salu   scalar ALU (s_cmp, s_and, s_mov...) branch s_branch s_cbranch*
    smem      s_load* s_store*       vmem   global_* buffer_* flat_* scratch_*  other  s_waitcnt s_nop s_sleep s_endpgm ...
SQ_INSTS_VALU counts fp64 + vcmp + vmov + lane + valu.
"""
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the disassembly itself (code objects out of the library, one kernel's instructions, their classes) is shared with the
# lints smartpy_amd.build runs on every library it links
from smartpy_amd.isa_lint import OBJDUMP, FP64, classify, parse, Disassembly     # noqa: E402,F401

VALU_CLASSES = ('fp64', 'vcmp', 'vmov', 'lane', 'valu')


def disassemble(lib, kernel):
    """-> (address of the kernel's first instruction, its mangled name, llvm-objdump's text of it)"""
    return Disassembly(lib).raw(kernel)


def blocks_and_loops(insts):
    addr_ix = {x['addr']: i for i, x in enumerate(insts)}
    leaders = {0}
    for i, x in enumerate(insts):
        if x['cls'] == 'branch':
            if i + 1 < len(insts):
                leaders.add(i + 1)
            if x['target'] in addr_ix:
                leaders.add(addr_ix[x['target']])
        if x['op'].startswith('s_endpgm') and i + 1 < len(insts):
            leaders.add(i + 1)
    order = sorted(leaders)
    blocks = []
    for b, lo in enumerate(order):
        hi = order[b + 1] if b + 1 < len(order) else len(insts)
        blocks.append({'id': b, 'lo': lo, 'hi': hi})
    block_of = {}
    for b in blocks:
        for i in range(b['lo'], b['hi']):
            block_of[i] = b['id']
    # back edges: a branch to an address at or before itself
    loops = []
    for i, x in enumerate(insts):
        if x['cls'] == 'branch' and x['target'] in addr_ix and addr_ix[x['target']] <= i:
            loops.append({'head': addr_ix[x['target']], 'tail': i})
    # loops that share a head are one loop (several latches); nest by containment of [head, tail]
    merged = {}
    for lp in loops:
        merged[lp['head']] = max(merged.get(lp['head'], 0), lp['tail'])
    loops = [{'id': k, 'head': h, 'tail': t} for k, (h, t) in enumerate(sorted(merged.items()))]
    for b in blocks:
        b['loops'] = [lp['id'] for lp in loops if lp['head'] <= b['lo'] and b['hi'] - 1 <= lp['tail']]
        b['hist'] = dict(Counter(insts[i]['cls'] for i in range(b['lo'], b['hi'])))
        b['ops'] = dict(Counter(insts[i]['op'].replace('_e32', '').replace('_e64', '') for i in range(b['lo'], b['hi'])))
    for lp in loops:
        lp['depth'] = sum(1 for o in loops if o['head'] <= lp['head'] and lp['tail'] <= o['tail'])
        lp['n_insts'] = lp['tail'] - lp['head'] + 1
        lp['hist'] = dict(Counter(insts[i]['cls'] for i in range(lp['head'], lp['tail'] + 1)))
    return blocks, loops, block_of


def fmt_hist(h):
    valu = sum(h.get(c, 0) for c in VALU_CLASSES)
    return 'VALU %d (fp64 %d, vcmp %d, vmov %d, lane %d, other %d) | SALU %d | branch %d | SMEM %d | VMEM %d' % (
        valu, h.get('fp64', 0), h.get('vcmp', 0), h.get('vmov', 0), h.get('lane', 0), h.get('valu', 0),
        h.get('salu', 0), h.get('branch', 0), h.get('smem', 0), h.get('vmem', 0))


def main():
    kernel, out = sys.argv[1], sys.argv[2]
    rest = [x for x in sys.argv[3:] if not x.startswith('--')]
    lib = rest[0] if rest else os.path.join(ROOT, 'smartpy_amd', 'csrc', 'libsmart_amd.so')
    start, symbol, body = disassemble(lib, kernel)
    insts = parse(start, body)
    blocks, loops, block_of = blocks_and_loops(insts)
    lines = ['; %s -- llvm-objdump -d of the gfx950 code object in %s' % (symbol, os.path.relpath(lib, ROOT)),
             '; %d instructions, %d basic blocks, %d loops (tools/isa_report.py)' % (len(insts), len(blocks), len(loops)),
             ';']
    for lp in sorted(loops, key=lambda l: l['head']):
        lines.append('; loop L%d depth %d: %#x .. %#x, %d instructions: %s' % (
            lp['id'], lp['depth'], insts[lp['head']]['addr'], insts[lp['tail']]['addr'], lp['n_insts'],
            fmt_hist(lp['hist'])))
    lines.append(';')
    heads = {lp['head']: lp for lp in loops}
    tails = {}
    for lp in loops:
        tails.setdefault(lp['tail'], []).append(lp)
    # --hot: list only the loops that are mostly fp64 arithmetic (the time-loop bodies) in full; the rest of the kernel
    # (set-up, hand-over, reports, the other variants' code) one line per block
    hot = {lp['id'] for lp in loops if 60 <= lp['n_insts'] <= 2800 and lp['depth'] >= (2 if len(loops) > 20 else 1) and
           lp['hist'].get('fp64', 0) + lp['hist'].get('valu', 0) >= 0.6 * lp['n_insts']} if '--hot' in sys.argv else None
    for b in blocks:
        if hot is not None and not (set(b['loops']) & hot):
            lines.append(';; (block B%d at %#x, %d instructions, not in a hot loop: %s)' % (
                b['id'], insts[b['lo']]['addr'], b['hi'] - b['lo'], fmt_hist(b['hist'])))
            continue
        if b['loops']:
            lines.append(';; ---- block B%d [loops %s] %s' % (b['id'], ','.join('L%d' % k for k in b['loops']),
                                                              fmt_hist(b['hist'])))
        for i in range(b['lo'], b['hi']):
            x = insts[i]
            if i in heads:
                lines.append(';; ==== LOOP L%d BEGIN (depth %d)' % (heads[i]['id'], heads[i]['depth']))
            tgt = ''
            if x['target'] is not None:
                j = next((k for k, y in enumerate(insts) if y['addr'] == x['target']), None)
                tgt = '    ; -> %#x (B%s)' % (x['target'], block_of.get(j, '?'))
            lines.append('%08x  %-18s %s%s' % (x['addr'], x['op'], x['args'], tgt))
            for lp in tails.get(i, []):
                lines.append(';; ==== LOOP L%d END' % lp['id'])
    with open(out + '.s', 'w') as fh:
        fh.write('\n'.join(lines) + '\n')
    # (one block, one loop per line: the tables of the kernels with several instances of their asm run to thousands of rows)
    block_rows = [{k: b[k] for k in ('id', 'loops', 'hist')} | {
        'addr': insts[b['lo']]['addr'], 'n': b['hi'] - b['lo'], 'ends_with': insts[b['hi'] - 1]['op'],
        'target': insts[b['hi'] - 1]['target']} for b in blocks]
    loop_rows = [{k: lp[k] for k in ('id', 'depth', 'n_insts', 'hist')} | {
        'head_addr': insts[lp['head']]['addr'], 'tail_addr': insts[lp['tail']]['addr']} for lp in loops]
    with open(out + '.json', 'w') as fh:
        fh.write('{"kernel": %s, "n_insts": %d,\n "blocks": [\n  %s\n ],\n "loops": [\n  %s\n ]}\n' % (
            json.dumps(symbol), len(insts), ',\n  '.join(json.dumps(r) for r in block_rows),
            ',\n  '.join(json.dumps(r) for r in loop_rows)))
    print('\n'.join(lines[:2]))
    for lp in sorted(loops, key=lambda l: -l['hist'].get('fp64', 0))[:6]:
        print('  L%d depth %d, %d instructions: %s' % (lp['id'], lp['depth'], lp['n_insts'], fmt_hist(lp['hist'])))


if __name__ == '__main__':
    main()
