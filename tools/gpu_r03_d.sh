#!/bin/bash
# round 3, fourth GPU pass: the interval-streaming arm loop (ping-pong buffers) against the flat chunk loop; bits; suite
export TMPDIR=/tmp
mkdir -p gpurun_out
for rep in 1 2 3; do
  for so in default ool0 flatloop oldsteps; do
    if [ "$so" = default ]; then unset SMART_AMD_LIB; else export SMART_AMD_LIB=$PWD/smartpy_amd/csrc/libsmart_amd_$so.so; fi
    echo -n "$so: "; python tools/debug/flat_only.py 100000 4 2>/dev/null | tail -2 | tr '\n' ' '; echo
  done
done 2>&1 | tee gpurun_out/ab_stream.log
for so in default ool0 flatloop; do
  if [ "$so" = default ]; then unset SMART_AMD_LIB; else export SMART_AMD_LIB=$PWD/smartpy_amd/csrc/libsmart_amd_$so.so; fi
  echo -n "1e6 $so: "; python tools/debug/flat_only.py 1000000 3 2>/dev/null | tail -2 | tr '\n' ' '; echo
done 2>&1 | tee -a gpurun_out/ab_stream.log
unset SMART_AMD_LIB
( SMART_AMD_LIB=$PWD/smartpy_amd/csrc/libsmart_amd_oldsteps.so timeout 900 python tools/debug/steps_bits.py dump /tmp/steps_old.npz
  timeout 900 python tools/debug/steps_bits.py dump /tmp/steps_new.npz
  python tools/debug/steps_bits.py compare /tmp/steps_old.npz /tmp/steps_new.npz ) > gpurun_out/steps_bits.log 2>&1
tail -2 gpurun_out/steps_bits.log
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_d.log 2>&1; tail -3 gpurun_out/pytest_d.log
mkdir -p gpurun_out/prof_r03_flat_d
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_BRANCH --kernel-trace --output-format csv -d gpurun_out/prof_r03_flat_d/pmc_SQ_WAVES -o pmc -- python3 tools/debug/flat_only.py 100000 6 > gpurun_out/prof_r03_flat_d/pmc.log 2>&1
python - <<'PY'
import csv,glob
from collections import defaultdict
d=defaultdict(list)
for p in glob.glob('gpurun_out/prof_r03_flat_d/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(p)):
        if 'smart_fast_steps' in r['Kernel_Name']: d[r['Counter_Name']].append(float(r['Counter_Value']))
ws=1563*105192
for k,v in sorted(d.items()): print(k, sum(v)/len(v), 'per wave-step %.3f'%(sum(v)/len(v)/ws))
PY
python bench.py --steps 6 --warmup 2 > gpurun_out/bench_d.log 2>&1; tail -1 gpurun_out/bench_d.log | cut -c1-1500
