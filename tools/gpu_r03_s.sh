#!/bin/bash
# round 3, nineteenth GPU pass: the ill-conditioned rows' kernel with its SIMD to itself (smart_fast_illcond_solo:
# register allocation 184 + 256 of 512) against the shared one, on the daily ensemble; per-kernel times
export TMPDIR=/tmp
mkdir -p gpurun_out
for rep in 1 2 3; do for solo in 1 0; do
  export SMART_ILLCOND_SOLO=$solo
  echo -n "SMART_ILLCOND_SOLO=$solo: config 2 "; python bench.py --config 2 --steps 20 --warmup 5 --no-cpu-baseline --no-flat --no-strong 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms/step  launch %.3f  %s' % (d['ms_per_step'], d['roofline']['launch_ms'], d['roofline']['kernel']))"
  rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o t -- python3 bench.py --config 2 --steps 20 --warmup 5 --no-cpu-baseline --no-flat --no-strong > /dev/null 2>&1
  python3 - <<'PY'
import csv, glob
for p in glob.glob('/tmp/kt/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        if 'smart_fast' in r['Name']: print('    %-30s avg %.3f ms  min %.3f' % (r['Name'].split('(')[0], float(r['AverageNs'])/1e6, float(r['MinNs'])/1e6))
PY
done; done 2>&1 | tee gpurun_out/ab_illcond_solo.log
unset SMART_ILLCOND_SOLO
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config2 or illcond or ill_conditioned or daily" 2>&1 | tail -2
