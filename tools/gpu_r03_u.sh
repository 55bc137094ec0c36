#!/bin/bash
# round 3, twenty-first GPU pass: the literal chain's evaporation cascade and the hand-down of the filling cascade
# without their selects (min forms, the same bits) in the reciprocal path.  base = the tree before
export TMPDIR=/tmp
mkdir -p gpurun_out
C=smartpy_amd/csrc
python tools/debug/recip_bits.py 2>&1 | tail -3 | tee gpurun_out/recip_bits_min.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -m gpu -x -q 2>&1 | tail -2
for rep in 1 2 3; do for so in default base; do
  if [ "$so" = default ]; then unset SMART_AMD_LIB; else export SMART_AMD_LIB=$PWD/$C/libsmart_amd_$so.so; fi
  echo -n "$so: config 2 "; python bench.py --config 2 --steps 20 --warmup 5 --no-cpu-baseline --no-flat --no-strong 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms/step  launch %.3f' % (d['ms_per_step'], d['roofline']['launch_ms']))"
  rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o t -- python3 bench.py --config 2 --steps 20 --warmup 5 --no-cpu-baseline --no-flat --no-strong > /dev/null 2>&1
  python3 - <<'PY'
import csv, glob
for p in glob.glob('/tmp/kt/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        if 'smart_fast_illcond' in r['Name']: print('    %-28s avg %.3f ms  min %.3f' % (r['Name'].split('(')[0], float(r['AverageNs'])/1e6, float(r['MinNs'])/1e6))
PY
  echo -n "$so: literal mode (config 3, 1e5 samples) "; python bench.py --math literal --steps 2 --warmup 1 --no-cpu-baseline --no-flat --no-strong 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms' % d['ms_per_step'])"
done; done 2>&1 | tee gpurun_out/ab_literal_min.log
