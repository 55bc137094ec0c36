#!/usr/bin/env python3
"""One bench.py JSON line on stdin -> a few lines a human reads (tools/gpu_round.sh bench, gpu_final_check.sh)."""
import json
import sys

d = json.loads(sys.stdin.read())
r = d['roofline']
print('value %.4g  ms/step %.3f  launch %.3f ms  frac %s useful %s held %s traffic %s' % (
    d['value'], d['ms_per_step'], r['launch_ms'], r['frac'], r['useful_frac'], r['frac_at_held_clock'], r['traffic']))
print('kernel:', r['kernel'])
print('pmc:', (r.get('pmc_source') or '')[:110])
for k in ('flat_forcing', 'runs_of_6', 'objectives_only', 'raw_gap24', 'gap1'):
    if k in d:
        print('%-16s %8.3f ms  %s' % (k, d[k]['launch_ms'], d[k]['kernel']))
if 'strong_1e6' in d:
    print('%-16s %8.3f ms  %s' % ('strong_1e6', d['strong_1e6']['ms_per_step'], d['strong_1e6']['kernel']))
if 'parity' in d:
    print('parity', d['parity']['max_rel_discharge'], 'cpu %.4g on %d cores' % (d['cpu_baseline']['value'],
                                                                              d['cpu_baseline']['cores']))
print('ranks:', d['ranks']['backend'], d['ranks']['world_size'], d['ranks'].get('rccl_version'))
