export TMPDIR=/tmp
C=smartpy_amd/csrc
for rep in 1 2 3; do for so in default nn base; do
  if [ "$so" = default ]; then unset SMART_AMD_LIB; else export SMART_AMD_LIB=$PWD/$C/libsmart_amd_$so.so; fi
  rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o t -- python3 bench.py --config 2 --steps 20 --warmup 5 --no-cpu-baseline --no-flat --no-strong > /dev/null 2>&1
  python3 - <<PY
import csv, glob
for p in glob.glob('/tmp/kt/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        if 'smart_fast_illcond' in r['Name']: print('$so    %-28s avg %.3f ms  min %.3f' % (r['Name'].split('(')[0], float(r['AverageNs'])/1e6, float(r['MinNs'])/1e6))
PY
done; done
