#!/usr/bin/env python3
"""Condense the rocprofv3 CSVs written by tools/profile.sh into a small markdown + json summary.
usage: python tools/summarize_profile.py gpurun_out/prof_<tag> profiles/<name> [workload-key]
With a workload key (bench.py prints it as roofline.pmc_key) the per-launch figures bench.py quotes -- HBM bytes,
vector-instruction count, held clock -- are also entered into profiles/traffic_latest.json under that key."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
workload = sys.argv[3] if len(sys.argv) > 3 else None


def is_main(name):
    """the time-loop kernels: smart_fast_* (one per variant) and smart_ensemble_literal"""
    return 'smart_fast_' in name or 'smart_ensemble' in name

out = {'source': src}
# bench.py's untimed warm-up launches come first: they are left out of every per-launch mean below
n_warm = 0
args_file = os.path.join(src, 'args.txt')
if os.path.exists(args_file):
    words = open(args_file).read().split()
    out['bench_args'] = ' '.join(words)
    if '--warmup' in words:
        n_warm = int(words[words.index('--warmup') + 1])


def dominant_first(kernels):
    """the time-loop kernels of the call, the one with the largest total time in the plain kernel trace first"""
    total = {r['kernel']: r['total_ms'] for r in out.get('kernel_stats', [])}
    return sorted(kernels, key=lambda k: -total.get(k, 0.0))


def fp64_share(kernel, measured_valu=None, workload=''):
    """share of the vector instructions that are fp64 arithmetic: the model's figure (a lower bound: hipcc's code
    around the asm loops is counted on both sides of its branches); for the workload the model describes (the headline
    run) the fp64 instructions of the wet steps over the MEASURED vector instructions"""
    name = kernel.split('::')[-1]
    found = sorted(glob.glob(os.path.join(os.path.dirname(dst) or '.', 'r0?_isa_model_%s.json'
                                          % name.replace('smart_fast_', ''))))
    if not found:
        return None
    model = json.load(open(found[-1]))              # the latest round's
    if measured_valu and workload.startswith('config3:runs_per_gpu=100000'):
        if 'fp64_per_wave_step' in model:          # round 4: the glue's arithmetic included
            return model['fp64_per_wave_step'] * model['wave_steps'] / measured_valu
        if 'fp64_in_wet_steps_per_wave_step' in model:
            return model['fp64_in_wet_steps_per_wave_step'] * model['wave_steps'] / measured_valu
    share = model.get('fp64_share_of_valu')
    return min(share) if isinstance(share, list) else share


def timed(values):
    """the per-dispatch values of the timed steps (the warm-up dispatches dropped, when there are enough left)"""
    return values[n_warm:] if len(values) > n_warm else values

lines = ['# rocprofv3 summary (%s)' % os.path.basename(src), '']

# ---- kernel trace: per-kernel stats ----------------------------------------------------------------
stats = glob.glob(os.path.join(src, 'trace', '**', '*kernel_stats.csv'), recursive=True)
if stats:
    lines += ['## `rocprofv3 --kernel-trace --stats` kernel stats', '', '| kernel | calls | total ms | avg ms | min ms | max ms | % |',
              '|---|---|---|---|---|---|---|']
    out['kernel_stats'] = []
    with open(stats[0]) as f:
        for row in csv.DictReader(f):
            name = row['Name'].split('(')[0]
            rec = {'kernel': name, 'calls': int(row['Calls']), 'total_ms': float(row['TotalDurationNs']) / 1e6,
                   'avg_ms': float(row['AverageNs']) / 1e6, 'min_ms': float(row['MinNs']) / 1e6,
                   'max_ms': float(row['MaxNs']) / 1e6, 'pct': float(row['Percentage'])}
            out['kernel_stats'].append(rec)
            lines.append('| `%s` | %d | %.3f | %.4f | %.4f | %.4f | %.2f |' % (
                name[:70], rec['calls'], rec['total_ms'], rec['avg_ms'], rec['min_ms'], rec['max_ms'], rec['pct']))
    lines.append('')
trace = glob.glob(os.path.join(src, 'trace', '**', '*kernel_trace.csv'), recursive=True)
if trace:
    with open(trace[0]) as f:
        rows = [r for r in csv.DictReader(f) if is_main(r['Kernel_Name'])]
    if rows:
        # the dominant kernel of the call: of the time-loop kernels, the one with the largest total time (a daily
        # ensemble runs three side by side: their dispatches must not be averaged together)
        top = dominant_first(sorted({r['Kernel_Name'].split('(')[0] for r in rows}))[0]
        rows = [r for r in rows if r['Kernel_Name'].split('(')[0] == top]
        out['dominant_kernel'] = top
        gmax = max(int(r['Grid_Size_X']) for r in rows)
        rows.sort(key=lambda r: int(r['Start_Timestamp']))
        every = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6 for r in rows if int(r['Grid_Size_X']) == gmax]
        full = timed(every)
        out['full_size_dispatch_ms'] = {'n': len(full), 'avg': sum(full) / len(full), 'min': min(full), 'max': max(full),
                                        'grid_x': gmax, 'warmup_dispatches_dropped': len(every) - len(full),
                                        'warmup_ms': every[:len(every) - len(full)]}
        lines += ['## full-size dispatches of the dominant kernel `%s` (grid %d threads; `%s`)' % (top, gmax, out.get('bench_args', '')),
                  '', '- the %d timed steps: avg = %.4f ms, min = %.4f ms, max = %.4f ms' % (len(full), sum(full) / len(full), min(full), max(full)),
                  '- the %d warm-up launches before them (clock still ramping): %s ms' % (
                      len(every) - len(full), ', '.join('%.3f' % v for v in every[:len(every) - len(full)])), '']
        r = [r for r in rows if int(r['Grid_Size_X']) == gmax][-1]
        keep = {k: r[k] for k in r if k in ('Kernel_Name', 'VGPR_Count', 'Accum_VGPR_Count', 'SGPR_Count', 'LDS_Block_Size',
                                            'Scratch_Size', 'Workgroup_Size', 'Grid_Size', 'Workgroup_Size_X', 'Grid_Size_X')}
        out['dispatch'] = keep
        lines += ['## dispatch of the dominant kernel', '', '```', json.dumps(keep, indent=1), '```', '']

# ---- PMC passes ------------------------------------------------------------------------------------
pmc = defaultdict(lambda: defaultdict(list))
clock = defaultdict(list)
allrows = []
order = {}
for path in glob.glob(os.path.join(src, 'pmc_*', '**', '*counter_collection.csv'), recursive=True):
    with open(path) as f:
        allrows += list(csv.DictReader(f))
# the bench also launches the time-loop kernel once with N = 1 (the synthetic "truth" run): keep only the
# full-size dispatches of the dominant kernel, and drop torch's / the runtime's helper kernels
big = max([int(r['Grid_Size']) for r in allrows if is_main(r['Kernel_Name'])] or [0])
for row in allrows:
    if 'smart' not in row['Kernel_Name']:
        continue
    if is_main(row['Kernel_Name']) and int(row['Grid_Size']) != big:
        continue
    pmc[row['Kernel_Name'].split('(')[0]][row['Counter_Name']].append((int(row['Dispatch_Id']), float(row['Counter_Value'])))
    if row['Counter_Name'] == 'GRBM_GUI_ACTIVE':     # the dispatch's own duration in the pass that counted its cycles
        secs = (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e9
        clock[row['Kernel_Name'].split('(')[0]].append((int(row['Dispatch_Id']), float(row['Counter_Value']) / 8.0 / secs))
if pmc:
    lines += ['## PMC counters (one `--pmc` pass each; value = mean per dispatch)', '', '| kernel | counter | mean per dispatch | dispatches |', '|---|---|---|---|']
    out['pmc'] = {}
    for k in sorted(pmc):
        out['pmc'][k] = {}
        for c in sorted(pmc[k]):
            v = [x[1] for x in sorted(pmc[k][c])]
            v = timed(v) if is_main(k) else v
            out['pmc'][k][c] = sum(v) / len(v)
            lines.append('| `%s` | %s | %.6g | %d |' % (k[:60], c, sum(v) / len(v), len(v)))
    lines.append('')
    main = dominant_first([k for k in pmc if is_main(k)])
    if main:
        # a call may run several time-loop kernels side by side (a daily ensemble: plain + stiff + illcond); its HBM
        # traffic is the sum over them, everything else is reported per kernel
        c = {n: sum(out['pmc'][k].get(n, 0.0) for k in main) for n in ('FETCH_SIZE', 'WRITE_SIZE')
             if all(n in out['pmc'][k] for k in main)}
        if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
            # MI355X_MICROARCH.md, HBM: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 128-B requests
            # as 64 B for wide coalesced streaming reads (x2).  This kernel's reads are scalar loads + 8-B/lane rows,
            # not 16-B/lane streams, so the x2 is an upper bound; both figures are given.
            rd, wr = c['FETCH_SIZE'] * 1024, c['WRITE_SIZE'] * 1024
            out['hbm_bytes_per_launch'] = 2 * rd + wr
            out['hbm_bytes_per_launch_uncorrected'] = rd + wr
            out['hbm_read_bytes_raw'] = rd
            out['hbm_write_bytes'] = wr
            lines += ['## HBM traffic of the dominant kernel, per launch', '',
                      '- FETCH_SIZE = %.4g KiB -> %.4g MB raw, %.4g MB with the gfx950 x2 correction' % (c['FETCH_SIZE'], rd / 1e6, 2 * rd / 1e6),
                      '- WRITE_SIZE = %.4g KiB -> %.4g MB' % (c['WRITE_SIZE'], wr / 1e6),
                      '- traffic (corrected) = %.4g MB per launch' % ((2 * rd + wr) / 1e6), '']
hash_file = os.path.join(src, 'source_hash.txt')
if os.path.exists(hash_file):
    out['source_hash'] = open(hash_file).read().strip()
    lines += ['kernel sources at profile time: sha256[:16] = `%s` (bench.kernel_source_hash)' % out['source_hash'], '']
main = dominant_first([k for k in out.get('pmc', {}) if is_main(k)])
if main and clock:
    # GRBM_GUI_ACTIVE counts busy cycles of every XCD (8); over the duration of THE SAME dispatch in the pass that
    # counted them, that is the engine clock the chip held.  Per kernel: under --pmc the dispatches of a call run one
    # after the other, so each kernel's counters and duration are its own (round 2 divided one kernel's cycles by the
    # duration of another that runs beside it in a normal launch, and printed 4.2 GHz).
    out['per_kernel'] = {}
    lines += ['## clock held and issue-slot fraction, per time-loop kernel', '',
              '| kernel | GRBM_GUI_ACTIVE / 8 / own duration | SQ_INSTS_VALU x 4 / (1,024 SIMDs x GRBM_GUI_ACTIVE / 8) | SALU per VALU |',
              '|---|---|---|---|']
    for k in main:
        c0 = out['pmc'][k]
        hz = [x[1] for x in sorted(clock.get(k, []))]
        hz = timed(hz)
        rec = {'held_clock_hz': sum(hz) / len(hz) if hz else None}
        if 'SQ_INSTS_VALU' in c0 and 'GRBM_GUI_ACTIVE' in c0:
            rec['issue_frac_at_held_clock'] = c0['SQ_INSTS_VALU'] * 4.0 / (128 * c0['GRBM_GUI_ACTIVE'])
        if 'SQ_INSTS_VALU' in c0 and 'SQ_INSTS_SALU' in c0 and c0['SQ_INSTS_VALU'] > 0:
            rec['salu_per_valu'] = c0['SQ_INSTS_SALU'] / c0['SQ_INSTS_VALU']
        out['per_kernel'][k] = rec
        lines.append('| `%s` | %s | %s | %s |' % (
            k[:60], '%.3f GHz' % (rec['held_clock_hz'] / 1e9) if rec['held_clock_hz'] else '-',
            '%.3f' % rec['issue_frac_at_held_clock'] if 'issue_frac_at_held_clock' in rec else '-',
            '%.3f' % rec['salu_per_valu'] if 'salu_per_valu' in rec else '-'))
    out['held_clock_hz'] = out['per_kernel'][main[0]]['held_clock_hz']
    # fp64 arithmetic by instruction class, COUNTED (SQ_INSTS_VALU_{FMA,ADD,MUL}_F64: wave-level instructions; v_min /
    # v_max / v_ldexp / v_cmp on doubles are in none of them) -> the flops the kernel executes, 64 lanes per instruction
    if any('SQ_INSTS_VALU_FMA_F64' in out['pmc'][k] for k in main):
        lines += ['', '## fp64 arithmetic as the counters see it, per time-loop kernel and launch', '',
                  '| kernel | FMA_F64 | ADD_F64 | MUL_F64 | TRANS_F64 | their share of SQ_INSTS_VALU | flops executed (2 FMA + ADD + MUL) x 64 | TFLOP/s at the clock held (of 78.6 x clock / 2.4 GHz) |',
                  '|---|---|---|---|---|---|---|---|']
        for k in main:
            c0 = out['pmc'][k]
            if 'SQ_INSTS_VALU_FMA_F64' not in c0:
                continue
            fma, add, mul = c0['SQ_INSTS_VALU_FMA_F64'], c0['SQ_INSTS_VALU_ADD_F64'], c0['SQ_INSTS_VALU_MUL_F64']
            rec = out['per_kernel'][k]
            rec['fp64_fma_add_mul_insts'] = fma + add + mul
            rec['fp64_flops_per_launch'] = (2 * fma + add + mul) * 64.0
            if c0.get('SQ_INSTS_VALU'):
                rec['fp64_fma_add_mul_share_of_valu'] = (fma + add + mul) / c0['SQ_INSTS_VALU']
            tfl = frac = None
            if 'GRBM_GUI_ACTIVE' in c0 and rec.get('held_clock_hz'):
                secs = c0['GRBM_GUI_ACTIVE'] / 8.0 / rec['held_clock_hz']       # the dispatch's duration in the cycle pass
                tfl = rec['fp64_flops_per_launch'] / secs / 1e12
                frac = tfl / (78.6 * rec['held_clock_hz'] / 2.4e9)
                rec['fp64_tflops_at_held_clock'], rec['fp64_flop_frac_at_held_clock'] = tfl, frac
            lines.append('| `%s` | %.4g | %.4g | %.4g | %.3g | %s | %.4g | %s |' % (
                k[:60], fma, add, mul, c0.get('SQ_INSTS_VALU_TRANS_F64', 0.0),
                '%.3f' % rec['fp64_fma_add_mul_share_of_valu'] if 'fp64_fma_add_mul_share_of_valu' in rec else '-',
                rec['fp64_flops_per_launch'], '%.1f (%.3f)' % (tfl, frac) if tfl else '-'))
    lines += ['', '(dominant kernel of the call: `%s`)' % main[0], '']
if workload and main and 'hbm_bytes_per_launch' in out:
    tpath = os.path.join(os.path.dirname(dst) or '.', 'traffic_latest.json')
    table = json.load(open(tpath)) if os.path.exists(tpath) else {}
    if 'workloads' not in table:
        table = {'workloads': {}}
    c = out['pmc'][main[0]]        # the dominant kernel (largest total time)
    table['workloads'][workload] = {
        'kernel': main[0], 'source_hash': out.get('source_hash'),
        'hbm_bytes_per_launch': out['hbm_bytes_per_launch'],
        'hbm_bytes_per_launch_uncorrected': out['hbm_bytes_per_launch_uncorrected'],
        'fetch_bytes_raw': out['hbm_read_bytes_raw'], 'write_bytes': out['hbm_write_bytes'],
        'valu_insts_per_launch': c.get('SQ_INSTS_VALU'), 'salu_insts_per_launch': c.get('SQ_INSTS_SALU'),
        'held_clock_hz': out.get('held_clock_hz'),
        # vector instructions x 4 issue cycles over the shader cycles the chip was busy for (GRBM_GUI_ACTIVE / 8 XCDs x
        # 1,024 SIMDs): the issue-slot fraction at the clock the chip held, from counters of the profiled runs alone
        'issue_frac_at_held_clock': (c['SQ_INSTS_VALU'] * 4.0 / (1024 * c['GRBM_GUI_ACTIVE'] / 8.0)
                                     if 'SQ_INSTS_VALU' in c and 'GRBM_GUI_ACTIVE' in c else None),
        'avg_ms_kernel_trace': out.get('full_size_dispatch_ms', {}).get('avg'),
        # share of the vector instructions that are fp64 arithmetic (tools/isa_model.py: the compiler's assembly of the
        # hot loop weighed with this workload's path frequencies); null for kernels without a model
        'fp64_share_of_valu': fp64_share(main[0], c.get('SQ_INSTS_VALU'), workload),
        # counted, not modelled: FMA + ADD + MUL on doubles (min / max / ldexp are in none of the three counters), and
        # the flops they are (2 per FMA, 64 lanes per instruction)
        'fp64_fma_add_mul_insts_per_launch': out['per_kernel'][main[0]].get('fp64_fma_add_mul_insts'),
        'fp64_flops_per_launch': out['per_kernel'][main[0]].get('fp64_flops_per_launch'),
        'source': dst + '.md: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU / GRBM_GUI_ACTIVE (separate '
                  'passes, tools/profile.sh) on the bench command of this workload; FETCH_SIZE doubled per '
                  'MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B), an upper bound here since the reads '
                  'are scalar loads and 8-B/lane rows'}
    json.dump(table, open(tpath, 'w'), indent=1)
with open(dst + '.md', 'w') as f:
    f.write('\n'.join(lines) + '\n')
with open(dst + '.json', 'w') as f:
    json.dump(out, f, indent=1)
print('\n'.join(lines))
