#!/usr/bin/env python3
"""A daily ensemble is THREE kernels side by side (one per arithmetic class of the rows), so its place against the issue
roof is the sum of their vector instructions over the SIMD cycles of the launch -- which tools/summarize_profile.py, made
for launches of one kernel, does not form.  From the raw rocprofv3 --pmc CSVs of tools/gpu_profile_r06.sh:

    python tools/daily_roofline.py gpurun_out/prof_r06_daily_1e6 profiles/r06_daily_1e6 [workload key]

appends a section to the .md (per kernel and summed: vector / scalar / branch instructions per launch, the launch's
duration = the longest kernel's in the kernel trace, the issue fraction at 2.4 GHz) and, with a key, enters the sum into
profiles/traffic_latest.json, where bench.leg_roofline finds it."""
import collections
import csv
import glob
import json
import os
import sys

src, dst = sys.argv[1], sys.argv[2]
key = sys.argv[3] if len(sys.argv) > 3 else None
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(src, 'pmc_*', '**', '*counter_collection.csv'), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row['Kernel_Name'].split('(')[0]
            if 'smart_fast_' in k:
                acc[k][row['Counter_Name']].append(float(row['Counter_Value']))
summary = json.load(open(dst + '.json'))
n_timed = summary['full_size_dispatch_ms']['n']
ms = summary['full_size_dispatch_ms']['avg']            # the longest kernel's dispatches = the launch
mean = {k: {c: sum(v[-n_timed:]) / len(v[-n_timed:]) for c, v in cs.items()} for k, cs in acc.items()}
valu = sum(m.get('SQ_INSTS_VALU', 0.0) for m in mean.values())
salu = sum(m.get('SQ_INSTS_SALU', 0.0) for m in mean.values())
branch = sum(m.get('SQ_INSTS_BRANCH', 0.0) for m in mean.values())
longest = summary['dominant_kernel']                     # (by time in the kernel trace: its dispatch spans the launch)
# (no fraction "at the clock held" here: the counter passes SERIALISE the three kernels, so a kernel's GRBM_GUI_ACTIVE is its
# own serial duration, not the launch's; the instruction counts do not care, and the duration is the kernel trace's, where
# the kernels do run side by side)
frac = valu * 4.0 / (1024 * 2.4e9 * ms * 1e-3)
held = None
lines = ['', '## the launch as a whole: its three kernels side by side against the issue roof (tools/daily_roofline.py)', '',
         '| kernel | vector instr. / launch | scalar | branches |', '|---|---|---|---|']
for k in sorted(mean, key=lambda k: -mean[k].get('SQ_INSTS_VALU', 0.0)):
    m = mean[k]
    lines.append('| `%s` | %.4g | %.4g | %.4g |' % (k, m.get('SQ_INSTS_VALU', 0), m.get('SQ_INSTS_SALU', 0), m.get('SQ_INSTS_BRANCH', 0)))
lines += ['| **sum** | **%.4g** | %.4g | %.4g |' % (valu, salu, branch), '',
          '- launch = the longest kernel of the kernel trace (`%s`): %.3f ms; vector instructions x 4 issue cycles over 1,024 '
          'SIMDs x that time at 2.4 GHz: **%.3f** (the counter passes serialise the kernels: counts from them, the time from '
          'the trace, where they run side by side)' % (longest, ms, frac), '']
text = open(dst + '.md').read()
marker = '## the launch as a whole'
if marker in text:
    text = text[:text.index(marker)].rstrip('\n') + '\n'
open(dst + '.md', 'w').write(text.rstrip('\n') + '\n' + '\n'.join(lines))
print('\n'.join(lines))
if key:
    path = os.path.join(os.path.dirname(dst) or '.', 'traffic_latest.json')
    table = json.load(open(path))
    table['workloads'][key] = {
        'kernel': ' + '.join(k.split('::')[-1] for k in sorted(mean, key=lambda k: -mean[k].get('SQ_INSTS_VALU', 0.0))),
        'valu_insts_per_launch': valu, 'salu_insts_per_launch': salu, 'issue_frac_at_held_clock': held,
        'avg_ms_kernel_trace': ms, 'source_hash': summary['source_hash'],
        'source': '%s.md: tools/daily_roofline.py (the three kernels of the launch summed)' % dst}
    with open(path, 'w') as fh:
        json.dump(table, fh, indent=1)
        fh.write('\n')
