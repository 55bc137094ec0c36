#!/bin/bash
# round 3, twenty-second GPU pass: the ill-conditioned chain on a stream of its own with the other kernels of the call
# one after the other beside it (SMART_CHAIN_ALONE=1, the default) against all of them side by side (=0)
export TMPDIR=/tmp
mkdir -p gpurun_out
for rep in 1 2 3; do for alone in 1 0; do
  export SMART_CHAIN_ALONE=$alone
  echo -n "SMART_CHAIN_ALONE=$alone: config 2 "; python bench.py --config 2 --steps 20 --warmup 5 --no-cpu-baseline --no-flat --no-strong 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms/step  launch %.3f' % (d['ms_per_step'], d['roofline']['launch_ms']))"
  rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o t -- python3 bench.py --config 2 --steps 20 --warmup 5 --no-cpu-baseline --no-flat --no-strong > /dev/null 2>&1
  python3 - <<'PY'
import csv, glob
for p in glob.glob('/tmp/kt/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        if 'smart_fast' in r['Name']: print('    %-28s avg %.3f ms  min %.3f' % (r['Name'].split('(')[0], float(r['AverageNs'])/1e6, float(r['MinNs'])/1e6))
PY
done; done 2>&1 | tee gpurun_out/ab_chain_alone.log
unset SMART_CHAIN_ALONE
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config2 or wild or stale or classes or variants" 2>&1 | tail -2
