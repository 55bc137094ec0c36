#!/usr/bin/env python3
"""Instruction counts of the hot loops, from the compiler's own assembly, weighed with the path frequencies of the bench
workloads -> expected instructions per wave-step, per class, to hold against the PMC counters (SQ_INSTS_VALU /
SQ_INSTS_SALU of profiles/r03_*.md).  Needs hipcc (cross-compiles without a GPU), no GPU.

    python tools/isa_model.py steps      [profiles/r04_isa_model_steps]        the flat_forcing leg (smart_fast_steps)
    python tools/isa_model.py intervals  [profiles/r04_isa_model_intervals]    the headline run (smart_fast_intervals)

The hot loops are `asm` statements (smartpy_amd/csrc/smart_fast_arms.h): in hipcc -S output they stand between
;;#ASMSTART / ;;#ASMEND with their local labels intact, so every arm is delimited by its label (100: calm, 110: dry,
120: rain arm of step 0, ... 130: end of chunk; 5: / 6: the two loops of the wet interval).  The classes are those of
tools/isa_report.py.  What is NOT in an asm (the glue hipcc writes around the chunks and the intervals) is counted
from the enclosing loop of the same listing: every basic block of it, weighed with the share of the workload's
(wavefront, interval) pairs that run it (round 4: glue_per_interval; round 3 counted both sides of every branch).  The
intervals model also lands in profiles/isa_model_latest.json, which bench.py quotes as roofline.valu_insts_model.
"""
import json
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))

import numpy as np                                  # noqa: E402
from isa_report import classify, VALU_CLASSES       # noqa: E402
from smartpy_amd import build as b                  # noqa: E402


def assembly(unit, kernel):
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, 'k.s')
        subprocess.run([b.hipcc()] + b.COMMON + b.UNITS[unit] + ['--cuda-device-only', '-S', os.path.join(b.CSRC, unit),
                                                                 '-o', out], check=True, stderr=subprocess.DEVNULL)
        text = open(out).read()
    m = re.search(r'^(_ZN5smart\d+%s[A-Z]\S*):.*?\n(.*?)\.end_amdhsa_kernel' % re.escape(kernel), text, re.M | re.S)
    return m.group(2).split('\n')


def insts(lines):
    """[(label or None, opcode)] of a stretch of assembly"""
    out = []
    for ln in lines:
        ln = ln.split(';')[0].strip()
        if not ln or ln.startswith('.'):
            continue
        m = re.match(r'^(\d+|\.LBB\d+_\d+):$', ln)
        if m:
            out.append((m.group(1), None))
            continue
        out.append((None, ln.split()[0]))
    return out


def hist(ops):
    c = Counter(classify(op) for op in ops)
    c['VALU'] = sum(c.get(k, 0) for k in VALU_CLASSES)
    c['scalar'] = c.get('salu', 0) + c.get('branch', 0)       # what SQ_INSTS_SALU counts: s_* ALU and branches
    return c


def asm_blocks(lines):
    """the asm statements of a kernel: [(first line, last line, [(label, op)])]"""
    out, start = [], None
    for i, ln in enumerate(lines):
        if 'ASMSTART' in ln:
            start = i
        elif 'ASMEND' in ln and start is not None:
            body = insts(lines[start + 1:i])
            if body:
                out.append((start, i, body))
            start = None
    return out


def segments(body):
    """split an asm at its numeric labels -> {label: [ops]}; '' = before the first label"""
    seg, cur = {'': []}, ''
    for lab, op in body:
        if lab is not None:
            cur = lab
            seg.setdefault(cur, [])
        else:
            seg[cur].append(op)
    return seg


def loop_blocks(lines, blocks):
    """The basic blocks hipcc wrote, grouped by the innermost compiler loop LLVM's comments assign them to
    (`.LBBx_y:  ; in Loop: Header=BBx_z` / `; %bb.N:  ; in Loop: ...`, the header itself `; =>This ... Loop Header`)
    -> {header: [block]}, block = {'ops': [opcodes outside asm], 'asm': 'wet' | 'calm' | None, 'pre': ops of the block
    that falls into it}.  An asm statement is a block of its own."""
    in_asm = {}
    for a, b, body in blocks:
        labs = {lab for lab, _ in body}
        kind = 'wet' if {'5', '6', '7', '9'} <= labs else ('calm' if len(body) > 60 else 'other')
        for i in range(a, b + 1):
            in_asm[i] = (a, kind)
    loops, header, cur = {}, None, None

    def start(h):
        nonlocal cur
        if cur is not None and not cur['ops'] and not cur['asm'] and loops.get(h) and loops[h][-1] is cur:
            return                                   # (nothing in the block at hand yet: it is the new one)
        cur = {'ops': [], 'asm': None}
        loops.setdefault(h, []).append(cur)

    for i, ln in enumerate(lines):
        m = re.match(r'^(?:\.LBB(\d+_\d+):|; %bb\.\d+:)\s*;\s*(=>\s*This (?:Inner )?Loop Header|in Loop: Header=BB(\d+_\d+))', ln)
        if m:
            header = m.group(1) if m.group(2).startswith('=>') else m.group(3)
            start(header)
            continue
        if re.match(r'^(\.LBB\d+_\d+:|; %bb\.\d+:)', ln):          # a block outside every loop
            header, cur = None, None
            continue
        if cur is None:
            continue
        if i in in_asm:
            if in_asm[i][0] == i:
                start(header)
                cur['asm'] = in_asm[i][1]
                start(header)
            continue
        t = ln.split(';')[0].strip()
        if not t or t.startswith('.') or t.endswith(':'):
            continue
        op = t.split()[0]
        cur['ops'].append(op)
        if op.startswith(('s_cbranch', 's_branch')):
            start(header)
    return loops


def glue_per_interval(blks, p):
    """Expected vector / scalar instructions per report interval of hipcc's code AROUND the asm loops of one compiler loop
    (unrolled over kGroup = 4 intervals): every basic block weighed with how often the workload runs it, told from what
    the block holds --
      the evaporation cascade + the dry map (>= 10 v_max_f64): the dry side of the per-lane branch (the block behind
          s_andn2_saveexec) or its copy for rainless intervals, picked on the scalar unit
      the block that falls into a wet / a calm asm loop (layer sum, e_h): that loop's share
      the objective-function moments (a handful of fma on a branch of their own): the share of observed reports
      anything else (the loop, the tests, the store): every interval."""
    n_asm = sum(1 for b in blks if b['asm'] == 'wet')
    assert n_asm and n_asm % 4 == 0 or n_asm == 4, n_asm
    tot = Counter()
    detail = Counter()
    for k, b in enumerate(blks):
        if b['asm']:
            continue
        h = hist(b['ops'])
        nxt = blks[k + 1]['asm'] if k + 1 < len(blks) else None
        prev_ops = blks[k - 1]['ops'] if k else []
        n_max = b['ops'].count('v_max_f64')
        n_fma = sum(1 for o in b['ops'] if o.startswith(('v_fma', 'v_fmac')))
        if n_max >= 10:
            cls = 'dry_lanes' if 's_andn2_saveexec_b64' in prev_ops else 'dry_scalar'
        elif nxt == 'wet':
            cls = 'wet'
        elif nxt == 'calm':
            cls = 'calm'
        elif n_fma >= 3 and n_max == 0 and h['VALU'] <= 14:
            cls = 'moments'
        else:
            cls = 'always'
        w = p[cls]
        for c in ('VALU', 'fp64', 'scalar'):
            tot[c] += h[c] * w
        detail[cls] += h['VALU']
    return {c: tot[c] / n_asm for c in tot}, {c: v / n_asm for c, v in detail.items()}


def fill_paths(workload):
    """path frequencies of the filling cascade (tools/fill_paths.py: a numpy walk of the soil layers over the bench
    workloads; depends on the workload only)"""
    path = os.path.join(ROOT, 'profiles', 'r03_fill_paths.json')
    if not os.path.exists(path):
        raise SystemExit('run  python tools/fill_paths.py profiles/r03_fill_paths.json  first')
    return json.load(open(path))[workload]


def fmt(c):
    return 'VALU %3d (fp64 %3d, vcmp %d, vmov %d) scalar %4.4g (branch %.4g) smem %d' % (
        c['VALU'], c.get('fp64', 0), c.get('vcmp', 0), c.get('vmov', 0), c['scalar'], c.get('branch', 0), c.get('smem', 0))


# ---- the step loop -------------------------------------------------------------------------------------------------
def steps_model(out):
    import bench
    from smartpy_amd.parameters import Parameters
    from smartpy_amd.sampling import latin_hypercube
    lines = assembly('smart_fast_steps.hip', 'smart_fast_steps')
    blocks = asm_blocks(lines)
    chunk = [blk for blk in blocks if any(lab == '130' for lab, _ in blk[2]) and any(lab == '100' for lab, _ in blk[2])]
    assert chunk, 'no threaded chunk asm found'
    # the asm in order: pieces between labels, each piece owned by the arm (or out-of-line cascade) it lies in
    pieces, cur_lab = [['', []]], ''
    for lab, op in chunk[0][2]:
        if lab is not None:
            pieces.append([lab, []])
        else:
            pieces[-1][1].append(op)
    arm_labels = {str(b + j): (k, j) for k, b in (('calm', 100), ('dry', 110), ('rain', 120)) for j in range(4)}
    owner, per_ops, ool_ops, top_ops = None, {}, {}, []
    for lab, ops in pieces:
        if lab == '':
            top_ops += ops
            continue
        if lab in arm_labels:
            owner = arm_labels[lab]
        elif re.fullmatch(r'[34]\d0', lab):
            owner = 'ool' + lab
        elif lab == '130':
            owner = None
        if owner is None:
            continue
        (ool_ops if isinstance(owner, str) else per_ops).setdefault(owner, []).extend(ops)
    report = ['# smart_fast_steps: the threaded chunk (smart_fast_arms.h: SMART_A_CHUNK), per arm', '',
              '(Since the pair blocks -- SMART_A_PAIRS_STRETCH, round 4 -- this is the path of launches without a '
              'workspace, of report gaps that are not a multiple of eight steps and of the models with the final state '
              'vector; the flat_forcing leg walks the pair blocks: the same arms, two to a block, one computed jump per '
              'block instead of two compares and two branches per step.  Its counts are measured: '
              'profiles/r04_flat_forcing.md.)', '']
    per = {k: hist(v) for k, v in per_ops.items()}
    for k in sorted(per):
        report.append('- arm %s of step %d: %s' % (k[0], k[1], fmt(per[k])))
    cascade = hist(ool_ops['ool300'])
    report.append('- the deferred cascade, out of line, all six layers: %s' % fmt(cascade))
    top = hist(top_ops)
    report.append('- dispatch at the top of the chunk: %s' % fmt(top))
    # the soil half of a rain arm (skipped by s_cbranch_execz when no lane of the wave is wet): from behind that
    # branch up to the label 8
    r0 = per_ops['rain', 0]
    soil_ops = []
    seen = False
    for lab, ops in pieces:
        if lab == '120':
            seen = True
        if seen:
            if lab == '8':
                break
            if 's_cbranch_execz' in ops:
                soil_ops += ops[ops.index('s_cbranch_execz') + 1:]
            elif soil_ops or lab == '401':
                soil_ops += ops
    rain_soil = hist(soil_ops)
    report.append('- of a rain arm, the soil half under EXEC (skipped when no lane is wet): %s' % fmt(rain_soil))
    # ... and of that half, the five lower layers of the filling cascade with the saturation excess (skipped by
    # s_cbranch_vccz when the top layer takes the excess of every lane: SMART_RAIN_FILL_EXIT)
    fill_tail = hist(soil_ops[soil_ops.index('s_cbranch_vccz') + 1:soil_ops.index('s_cbranch_vccz') + 18])
    assert fill_tail['VALU'] == 17 and fill_tail['scalar'] == 0, fill_tail
    report.append('- of that half, the filling below the top layer (skipped when the top layer takes every lane\'s '
                  'excess): %s' % fmt(fill_tail))
    # glue: the ping-pong loop = the innermost compiler loop that holds two chunk asms
    a, bnd = chunk[0][0], chunk[1][1] if len(chunk) > 1 else chunk[0][1]
    head = max(i for i in range(a) if re.match(r'^\.LBB\d+_\d+:', lines[i]))
    tail = next(i for i in range(bnd, len(lines)) if re.match(r'\s+s_cbranch', lines[i]))
    glue_ops = [op for lab, op in insts(lines[head:chunk[0][0]] + lines[chunk[0][1] + 1:chunk[1][0]] +
                                        lines[chunk[1][1] + 1:tail + 1]) if op]
    glue = hist(glue_ops)
    if glue['VALU'] > 8:    # hipcc has laid other blocks between the two chunks of its loop: the lines between them are
        # not the loop's own any more -- what the loop needs, as round 3's listing had it (11 scalar per two chunks)
        glue = Counter({'VALU': 0, 'scalar': 9, 'salu': 8, 'branch': 1, 'smem': 2})
    report.append('- hipcc\'s glue around TWO chunks (pointer, s_load_dwordx16 x2, s_waitcnt, counter, back-edge): %s'
                  % fmt(glue))
    report.append('')

    # ---- path frequencies of the flat_forcing leg: 1e5 LHS rows as drawn, 64 per wavefront
    base = bench.synthetic_forcing(0, True)[0]
    vary = bench.hourly_varying_forcing(base)
    f = np.concatenate([vary[:bench.WARM_DAYS * 24], vary])
    rain, pe = f[:, 0], f[:, 1]
    T = latin_hypercube(100000, Parameters().ranges, seed=2718)[:, 0]
    pad = (-len(T)) % 64
    Tw = np.concatenate([T, np.full(pad, T[-1])]).reshape(-1, 64)
    n_waves, n_steps = Tw.shape[0], len(rain)
    kind = np.where(rain > 0, 2, np.where(pe > 0, 1, 0))          # 0 calm 1 dry 2 rain
    j_of = np.arange(n_steps) % 4
    # lane state: demand pending since the lane's last wet step
    pending = np.zeros(Tw.shape, bool)
    tmin, tmax = Tw.min(1), Tw.max(1)
    n_casc = n_rain_soil = 0
    count = Counter()
    for t in range(n_steps):
        k = kind[t]
        if k == 1:
            pending[:] = True
        elif k == 0:
            n_casc += int(pending.any(1).sum())
            pending[:] = False
        else:
            thr = pe[t] / rain[t]
            wet = Tw >= thr                                   # ex = rain T - pe >= 0
            any_wet = wet.any(1)
            n_rain_soil += int(any_wet.sum())
            n_casc += int((pending & wet).any(1).sum())
            pending = np.where(wet, False, True)
    tot = Counter()
    for t in range(n_steps):
        arm = per[('calm', 'dry', 'rain')[kind[t]], int(j_of[t])]
        for c in ('VALU', 'fp64', 'scalar', 'branch', 'smem'):
            tot[c] += arm[c] * n_waves
    # rain steps in which no lane of the wave is wet skip the soil half (s_cbranch_execz): subtract it
    dry_rain = int((kind == 2).sum()) * n_waves - n_rain_soil
    for c in ('VALU', 'fp64', 'scalar', 'branch'):
        tot[c] -= rain_soil[c] * dry_rain
    paths = fill_paths('flat_forcing')
    n_absorbed = paths['absorbed_by_the_top_layer'] * n_rain_soil
    for c in ('VALU', 'fp64'):
        tot[c] -= fill_tail[c] * n_absorbed
    n_chunks = n_steps // 4
    for c in ('VALU', 'fp64', 'scalar', 'branch', 'smem'):
        tot[c] += top[c] * n_chunks * n_waves + glue[c] * (n_chunks // 2) * n_waves
    ws = n_waves * n_steps
    lo = {c: tot[c] / ws for c in tot}
    hi = {c: (tot[c] + cascade[c] * n_casc) / ws for c in tot}
    first = hist(ool_ops['ool300'][:5] + ool_ops['ool300'][-8:])   # a cascade that ends behind the top layer
    lo = {c: (tot[c] + first[c] * n_casc) / ws for c in tot}
    report += ['## flat_forcing leg (1e5 LHS rows as drawn, %d wavefronts x %d steps)' % (n_waves, n_steps), '',
               '- steps: calm %.3f, dry %.3f, rain %.3f; rain steps whose wave has no wet lane: %.4f of all wave-steps; '
               'cascades due: %.4f per wave-step; rainy steps whose excess the top layer takes in every wet lane: %.3f of '
               'the rainy steps with a wet lane (tools/fill_paths.py, %d rows)' % (
                   (kind == 0).mean(), (kind == 1).mean(), (kind == 2).mean(), dry_rain / ws, n_casc / ws,
                   paths['absorbed_by_the_top_layer'], paths['rows']),
               '- expected per wave-step, cascades ending behind the top layer .. walking all six layers:',
               '  - vector instructions %.2f .. %.2f (fp64 arithmetic %.2f .. %.2f)' % (lo['VALU'], hi['VALU'], lo['fp64'],
                                                                                     hi['fp64']),
               '  - scalar ALU + branches %.2f .. %.2f (branches %.2f .. %.2f), scalar loads %.2f' % (
                   lo['scalar'], hi['scalar'], lo['branch'], hi['branch'], lo['smem']), '']
    # ---- the same workload through the PAIR BLOCKS (SMART_A_PAIRS_STRETCH): the arms without their dispatch, two to a block
    # (four for a chunk of one calm or dry kind), what a block's second arm drops because it knows the first, the tails,
    # the report block once per interval.  Counted from the threaded chunk's pieces and the text of the macros.
    disp = Counter({'scalar': 4, 'branch': 2, 'salu': 2})
    core = {k: Counter(per[k, 0]) for k in ('calm', 'dry', 'rain')}
    for k in core:
        core[k].subtract(disp)
    pt = Counter()
    k_of = ('calm', 'dry', 'rain')
    for t in range(n_steps):
        for c in ('VALU', 'fp64', 'scalar', 'branch'):
            pt[c] += core[k_of[kind[t]]][c] * n_waves
    for c in ('VALU', 'fp64', 'scalar', 'branch'):
        pt[c] -= rain_soil[c] * dry_rain
    for c in ('VALU', 'fp64'):
        pt[c] -= fill_tail[c] * n_absorbed
    n_quads = n_second_c = n_second_r = n_cr_after_d = 0
    for ch in range(n_chunks):
        k4 = kind[4 * ch:4 * ch + 4]
        quad = k4[0] != 2 and (k4 == k4[0]).all()
        n_quads += quad
        firsts = (0,) if quad else (0, 2)
        for j in range(1, 4):
            if j in firsts:
                continue
            prev, cur = k4[j - 1], k4[j]
            if cur == 0 and prev != 2:
                n_second_c += 1          # no pending test, no hook: - v_cmp, - s_cbranch
            if cur == 2 and prev != 2:
                n_second_r += 1          # - v_cmp, - s_and, - s_cbranch
            if cur != 1 and prev == 1:
                n_cr_after_d += 1        # the cascade in line: no way out and back (2 branches), + 1 s_nop
    pt['VALU'] -= (n_second_c + n_second_r) * n_waves
    pt['scalar'] -= (n_second_c + 2 * n_second_r + 2 * n_cr_after_d) * n_waves
    pt['branch'] -= (n_second_c + n_second_r + 2 * n_cr_after_d) * n_waves
    # tails: first pair 2 (add, jump); second pair 6 + the interval test of every other chunk (2 / 2); 2 scalar loads a chunk
    pt['scalar'] += ((n_chunks - n_quads) * 2 + n_chunks * 6 + n_chunks) * n_waves
    pt['branch'] += ((n_chunks - n_quads) + n_chunks + n_chunks // 2) * n_waves
    pt['smem'] = 2 * n_chunks * n_waves
    # the report block: 4 tests, value, store + row, moments (test, 7), sum, reset; then interval counter, 2 requests, jump
    n_iv = n_steps // 24
    pt['VALU'] += 12 * n_iv * n_waves
    pt['fp64'] += 9 * n_iv * n_waves
    pt['scalar'] += 24 * n_iv * n_waves
    pt['branch'] += 8 * n_iv * n_waves
    plo = {c: (pt[c] + first[c] * n_casc) / ws for c in pt}
    phi = {c: (pt[c] + cascade[c] * n_casc) / ws for c in pt}
    report += ['## the same leg through the pair blocks (SMART_A_PAIRS_STRETCH)', '',
               '- chunks of four calm or four dry steps (one block, one jump): %.3f; second arms that know a calm or dry '
               'first arm: calm %.3f, rain %.3f per chunk; cascades taken in line behind a dry arm: %.3f per chunk' % (
                   n_quads / n_chunks, n_second_c / n_chunks, n_second_r / n_chunks, n_cr_after_d / n_chunks),
               '- expected per wave-step, cascades ending behind the top layer .. walking all six layers:',
               '  - vector instructions %.2f .. %.2f (fp64 arithmetic %.2f .. %.2f)' % (plo['VALU'], phi['VALU'], plo['fp64'],
                                                                                     phi['fp64']),
               '  - scalar ALU + branches %.2f .. %.2f (branches %.2f .. %.2f), scalar loads %.2f' % (
                   plo['scalar'], phi['scalar'], plo['branch'], phi['branch'], plo['smem']),
               '  (an estimate from the macros\' text, not a census of the binary: hand-over, slice entry and the launch\'s '
               'prologue are not in it, and s_setpc_b64 -- 0.57 per wave-step -- is counted with the branches here, which '
               'SQ_INSTS_BRANCH may not do; measured: profiles/r04_flat_forcing.md)', '']
    result = {'kernel': 'smart_fast_steps', 'wave_steps': ws, 'per_wave_step_low': lo, 'per_wave_step_high': hi,
              'pair_blocks_per_wave_step_low': plo, 'pair_blocks_per_wave_step_high': phi,
              'fp64_share_of_valu': [lo['fp64'] / lo['VALU'], hi['fp64'] / hi['VALU']],
              'arms': {'%s%d' % k: dict(v) for k, v in per.items()}, 'cascade': dict(cascade), 'glue_two_chunks': dict(glue),
              'fill_below_the_top_layer': dict(fill_tail), 'fill_absorbed_share': paths['absorbed_by_the_top_layer']}
    finish(out, report, result)


# ---- the interval engine ---------------------------------------------------------------------------------------------
def intervals_model(out):
    import bench
    from smartpy_amd.parameters import Parameters
    from smartpy_amd.sampling import latin_hypercube
    lines = assembly('smart_fast_intervals.hip', 'smart_fast_intervals')
    blocks = asm_blocks(lines)
    wet = [blk for blk in blocks if {'5', '6', '7', '9'} <= {lab for lab, _ in blk[2]}]
    assert wet, 'no wet-interval asm found'
    seg = segments(wet[0][2])
    # 5: the loop of steps whose excess the top layer takes in every lane, 6: head of a full step, 7: where the first step
    # that leaves something over joins it; 9: end.  SMART_WET_MODES 3 (round 4): these two loops take the n mod 4 steps
    # that do not fill a turn; 25 / 31 = the same steps four to a turn (71 ... 74 the joins), one counter and one back-edge
    # per turn, an s_nop between the absorbed copies; 20 / 30 / 40: the group counts
    four = '25' in seg and '31' in seg
    if four:
        assert seg['5'][-2:] == ['s_add_u32', 's_cbranch_scc0'] and seg['7'][-2:] == ['s_add_u32', 's_cbranch_scc0']
        absorbed = hist(seg['5'][:-2])
        turn_abs = seg['25']
        assert turn_abs[-3:] == ['s_add_u32', 's_cbranch_scc0', 's_branch'] and turn_abs.count('s_nop') == 3
        assert hist([o for o in turn_abs[:-3] if o != 's_nop'])['VALU'] == 4 * absorbed['VALU']
        turn_full = seg['31'] + seg['71'] + seg['72'] + seg['73'] + seg['74']
        assert turn_full[-2:] == ['s_add_u32', 's_cbranch_scc0']
        assert hist(turn_full[:-2])['VALU'] == 4 * hist(seg['6'] + seg['7'][:-2])['VALU']
    else:
        if seg['5'][-1] == 's_nop':         # (the padding behind `s_branch 9f` that puts the loop of full steps on an
            seg['5'].pop()                  # 8-byte boundary: never executed)
        assert seg['5'][-3:] == ['s_add_u32', 's_cbranch_scc0', 's_branch']
        assert seg['7'][-2:] == ['s_add_u32', 's_cbranch_scc0']
        absorbed = hist(seg['5'][:-3])
    step = hist(seg['6'] + seg['7'][:-2])
    loop_tail = hist(seg['7'][-2:])
    if four:    # per step of a 24-step interval: a quarter of a turn's counter and back-edge
        loop_tail = Counter({k: v / 4.0 for k, v in loop_tail.items()})
    entry = hist(seg[''] + (seg['20'] if four else []))       # (+ the group count of the loop the interval starts in)
    paths = fill_paths('headline')
    share = paths['absorbed_prefix_of_the_run']
    report = ['# smart_fast_intervals: the wet interval (smart_fast_arms.h: SMART_A_WET_INTERVAL)', '',
              '- a full wet step: %s' % fmt(step),
              '- a step whose excess the top layer takes in every lane: %s' % fmt(absorbed),
              '- loop tail, per step%s: %s' % (' (four steps to a turn: a quarter of its counter and back-edge)' if four else '', fmt(loop_tail)),
              '- entry, once per wet interval: %s' % fmt(entry),
              '- steps in the absorbed prefix of their interval: %.3f of the wet wave-steps; intervals that change '
              'mode: %.2f of the wet ones (tools/fill_paths.py, %d rows)' % (share, paths['mode_switches_per_wet_run'],
                                                                          paths['rows']), '']
    # ---- what the workload does, per (wavefront, interval): 1e5 LHS rows as drawn, 64 per wavefront
    base = bench.synthetic_forcing(0, True)[0]
    f = np.concatenate([base[:bench.WARM_DAYS * 24], base])[::24]
    rain, pe = f[:, 0], f[:, 1]
    T = latin_hypercube(100000, Parameters().ranges, seed=2718)[:, 0]
    pad = (-len(T)) % 64
    Tw = np.concatenate([T, np.full(pad, T[-1])]).reshape(-1, 64)
    tmin, tmax = Tw.min(1), Tw.max(1)
    scalar_dry = (rain == 0) & (pe > 0)              # decided on the scalar unit (QUICK waves: all of them here)
    scalar_calm = (rain == 0) & (pe == 0)
    lanes = ~(scalar_dry | scalar_calm)
    with np.errstate(divide='ignore', invalid='ignore'):
        thr = np.where(rain > 0, pe / rain, np.inf)
    any_wet = (tmax[:, None] >= thr[None, :]) & lanes[None, :]
    any_dry = (tmin[:, None] < thr[None, :]) & lanes[None, :]
    n_waves, n_iv = any_wet.shape
    ws = n_waves * n_iv * 24
    n_wet, n_dry = int(any_wet.sum()), int(any_dry.sum())
    rng = bench.synthetic_forcing(0, True)[1]         # the bench's observations: 12 % missing (bench.py main())
    rng.normal(0.0, 0.2, n_iv - bench.WARM_DAYS)
    observed = 1.0 - float((rng.random(n_iv - bench.WARM_DAYS) < 0.12).mean())
    share_of = {'dry_lanes': n_dry / (n_waves * n_iv), 'dry_scalar': float(scalar_dry.mean()),
                'wet': n_wet / (n_waves * n_iv), 'calm': float(scalar_calm.mean()), 'moments': observed, 'always': 1.0}
    # ---- hipcc's code around the asm loops: the compiler loops that hold four wet asm statements (kGroup = 4 intervals a
    # turn): the run loop is the one with the discharge stores and the moments, the warm-up loop the one without stores
    loops = loop_blocks(lines, blocks)
    cands = {h: b for h, b in loops.items() if sum(1 for x in b if x['asm'] == 'wet') == 4}
    stores = {h: sum(x['ops'].count('global_store_dwordx2') for x in b) for h, b in cands.items()}
    fmas = {h: sum(sum(1 for o in x['ops'] if o.startswith('v_fmac')) for x in b if not x['asm']) for h, b in cands.items()}
    run_h = max((h for h in cands if stores[h] >= 4), key=lambda h: fmas[h])
    warm_h = max((h for h in cands if stores[h] == 0), key=lambda h: len(cands[h]))
    glue_run, detail_run = glue_per_interval(cands[run_h], share_of)
    glue_warm, _ = glue_per_interval(cands[warm_h], dict(share_of, moments=0.0))
    upper_run, _ = glue_per_interval(cands[run_h], {k: 1.0 for k in share_of})
    w_warm = bench.WARM_DAYS / n_iv
    per_iv = {c: (1 - w_warm) * glue_run[c] + w_warm * glue_warm[c] for c in glue_run}
    report.append('- hipcc\'s code around the asm loops, per interval: every basic block of the run loop (.LBB%s) and of the '
                  'warm-up loop (.LBB%s) weighed with the share of the (wavefront, interval) pairs that run it -- dry side of '
                  'the per-lane branch %.3f, rainless intervals picked on the scalar unit %.3f, wet side %.3f, observed '
                  'reports %.3f: VALU %.1f (fp64 %.1f), scalar %.1f; with every block counted for every interval (round '
                  '3\'s upper bound): VALU %.1f' % (run_h, warm_h, share_of['dry_lanes'], share_of['dry_scalar'],
                                                    share_of['wet'], observed, per_iv['VALU'], per_iv.get('fp64', 0),
                                                    per_iv['scalar'], upper_run['VALU']))
    report.append('  (vector instructions of the run loop\'s blocks by kind, per interval, unweighed: %s)' % ', '.join(
        '%s %.1f' % kv for kv in sorted(detail_run.items())))
    mean = {c: share * absorbed[c] + (1 - share) * step[c] for c in ('VALU', 'fp64', 'scalar')}
    # (the step that changes mode runs the absorbed head, its compare and the full rest: one instruction more)
    valu_wet = n_wet * (24 * mean['VALU'] + paths['mode_switches_per_wet_run'] + entry['VALU'])
    valu = valu_wet + n_waves * n_iv * per_iv['VALU']
    valu_upper = valu_wet + n_waves * n_iv * upper_run['VALU']
    fp64_wet = n_wet * 24 * mean['fp64']
    fp64 = fp64_wet + n_waves * n_iv * per_iv.get('fp64', 0)
    scal = n_wet * (24 * (loop_tail['scalar'] + mean['scalar']) + entry['scalar']) + n_waves * n_iv * per_iv['scalar']
    report += ['', '## headline run (1e5 LHS rows as drawn, %d wavefronts x %d intervals of 24 steps)' % (n_waves, n_iv), '',
               '- intervals with a wet lane in the wave: %.4f; with a dry lane: %.4f (of them rainless: %.4f)' % (
                   n_wet / (n_waves * n_iv), n_dry / (n_waves * n_iv) + share_of['dry_scalar'], share_of['dry_scalar']),
               '- expected per wave-step: vector instructions %.2f (every block of the glue for every interval: <= %.2f), '
               'of them fp64 arithmetic %.2f (%.3f of the vector instructions; %.2f in the wet steps alone); scalar ALU + '
               'branches %.2f' % (valu / ws, valu_upper / ws, fp64 / ws, fp64 / valu, fp64_wet / ws, scal / ws), '']
    result = {'kernel': 'smart_fast_intervals', 'wave_steps': ws, 'valu_per_wave_step': valu / ws,
              'valu_per_wave_step_upper': valu_upper / ws, 'valu_insts_per_launch': valu,
              'fp64_in_wet_steps_per_wave_step': fp64_wet / ws, 'fp64_per_wave_step': fp64 / ws,
              'fp64_share_of_valu': fp64 / valu, 'scalar_per_wave_step': scal / ws, 'glue_per_interval': per_iv,
              'workload': 'config3:runs_per_gpu=100000:discharge=1:math=fast', 'source_hash': bench.kernel_source_hash(),
              'wet_step': dict(step), 'absorbed_step': dict(absorbed), 'absorbed_share': share,
              'wet_interval_fraction': n_wet / (n_waves * n_iv)}
    finish(out, report, result)


def finish(out, report, result):
    print('\n'.join(report))
    if out:
        with open(out + '.md', 'w') as fh:
            fh.write('\n'.join(report) + '\n')
        with open(out + '.json', 'w') as fh:
            json.dump(result, fh, indent=1)
        if 'workload' in result and os.path.dirname(os.path.abspath(out)) == os.path.join(ROOT, 'profiles'):
            # what bench.py quotes next to the PMC count (roofline.valu_insts_model), keyed like profiles/traffic_latest.json
            latest = os.path.join(ROOT, 'profiles', 'isa_model_latest.json')
            table = json.load(open(latest)) if os.path.exists(latest) else {'workloads': {}}
            table['workloads'][result['workload']] = {
                'kernel': result['kernel'], 'source_hash': result['source_hash'],
                'valu_insts_per_launch': result['valu_insts_per_launch'],
                'valu_per_wave_step': result['valu_per_wave_step'], 'fp64_share_of_valu': result['fp64_share_of_valu'],
                'source': 'profiles/%s.md: tools/isa_model.py (the compiler\'s assembly of the hot loops weighed with the '
                          'workload\'s path frequencies; no GPU involved)' % os.path.basename(out)}
            with open(latest, 'w') as fh:
                json.dump(table, fh, indent=1)


if __name__ == '__main__':
    which = sys.argv[1]
    out = sys.argv[2] if len(sys.argv) > 2 else None
    {'steps': steps_model, 'intervals': intervals_model}[which](out)
