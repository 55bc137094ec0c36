#!/bin/bash
# round 3, twentieth GPU pass: observation and deviation as ONE 16-byte scalar load per report (pairs in the
# workspace) instead of two dependent 8-byte ones.  base = the tree before (tools/build_variants.py)
export TMPDIR=/tmp
mkdir -p gpurun_out
C=smartpy_amd/csrc
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -m gpu -x -q 2>&1 | tail -2
bash tools/ab_variants.sh $C/libsmart_amd_base.so -- --no-strong 2>&1 | tee gpurun_out/ab_obs_pairs.log
for rep in 1 2; do for so in default base; do
  if [ "$so" = default ]; then unset SMART_AMD_LIB; else export SMART_AMD_LIB=$PWD/$C/libsmart_amd_$so.so; fi
  echo -n "$so: config 2 "; python bench.py --config 2 --steps 20 --warmup 5 --no-cpu-baseline --no-flat --no-strong 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms/step  launch %.3f' % (d['ms_per_step'], d['roofline']['launch_ms']))"
  rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o t -- python3 bench.py --config 2 --steps 20 --warmup 5 --no-cpu-baseline --no-flat --no-strong > /dev/null 2>&1
  python3 - <<'PY'
import csv, glob
for p in glob.glob('/tmp/kt/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        if 'smart_fast' in r['Name']: print('    %-28s avg %.3f ms  min %.3f' % (r['Name'].split('(')[0], float(r['AverageNs'])/1e6, float(r['MinNs'])/1e6))
PY
  echo -n "$so: flat 1e6 "; python tools/debug/flat_only.py 1000000 3 2>/dev/null | tail -2 | tr '\n' ' '; echo
done; done 2>&1 | tee -a gpurun_out/ab_obs_pairs.log
