#!/bin/bash
# round 3, second GPU pass: bits of the asm arms against the compiled loop; cascade in line / out of line; PMC of the flat leg
export TMPDIR=/tmp
mkdir -p gpurun_out
( SMART_AMD_LIB=$PWD/smartpy_amd/csrc/libsmart_amd_oldsteps.so timeout 900 python tools/debug/steps_bits.py dump /tmp/steps_old.npz
  timeout 900 python tools/debug/steps_bits.py dump /tmp/steps_new.npz
  python tools/debug/steps_bits.py compare /tmp/steps_old.npz /tmp/steps_new.npz ) > gpurun_out/steps_bits.log 2>&1
tail -3 gpurun_out/steps_bits.log
for rep in 1 2 3; do
  for so in default ool0 oldsteps; do
    if [ "$so" = default ]; then unset SMART_AMD_LIB; else export SMART_AMD_LIB=$PWD/smartpy_amd/csrc/libsmart_amd_$so.so; fi
    echo -n "$so: "; python tools/debug/flat_only.py 100000 4 | tail -2 | tr '\n' ' '; echo
  done
done 2>&1 | tee gpurun_out/ab_ool.log
unset SMART_AMD_LIB
echo -n "1e6: "; python tools/debug/flat_only.py 1000000 3 | tail -2 | tr '\n' ' '; echo
mkdir -p gpurun_out/prof_r03_flat_a
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_BRANCH --kernel-trace --output-format csv -d gpurun_out/prof_r03_flat_a/pmc_SQ_WAVES -o pmc -- python3 tools/debug/flat_only.py 100000 6 > gpurun_out/prof_r03_flat_a/pmc.log 2>&1
python - <<'PY'
import csv,glob
from collections import defaultdict
d=defaultdict(list)
for p in glob.glob('gpurun_out/prof_r03_flat_a/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(p)):
        if 'smart_fast_steps' in r['Kernel_Name']: d[r['Counter_Name']].append(float(r['Counter_Value']))
ws=1563*105192
for k,v in sorted(d.items()): print(k, sum(v)/len(v), 'per wave-step %.3f'%(sum(v)/len(v)/ws))
PY
