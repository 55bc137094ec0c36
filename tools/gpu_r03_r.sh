#!/bin/bash
# round 3, eighteenth GPU pass: the daily ensemble (config 2) with the next observation requested a step ahead in
# run_ensemble(): as scalar loads (roa1), as vector loads (roa2); per-kernel times from the kernel trace
export TMPDIR=/tmp
mkdir -p gpurun_out
C=smartpy_amd/csrc
for rep in 1 2; do for so in default roa1 roa2; do
  if [ "$so" = default ]; then unset SMART_AMD_LIB; else export SMART_AMD_LIB=$PWD/$C/libsmart_amd_$so.so; fi
  echo -n "$so: config 2 "; python bench.py --config 2 --steps 20 --warmup 5 --no-cpu-baseline --no-flat --no-strong 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms/step  launch %.3f' % (d['ms_per_step'], d['roofline']['launch_ms']))"
  rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o t -- python3 bench.py --config 2 --steps 20 --warmup 5 --no-cpu-baseline --no-flat --no-strong > /dev/null 2>&1
  python3 - <<'PY'
import csv, glob
for p in glob.glob('/tmp/kt/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        if 'smart_fast' in r['Name']: print('    %-28s avg %.3f ms  min %.3f' % (r['Name'].split('(')[0], float(r['AverageNs'])/1e6, float(r['MinNs'])/1e6))
PY
done; done 2>&1 | tee gpurun_out/ab_run_obs_ahead.log
