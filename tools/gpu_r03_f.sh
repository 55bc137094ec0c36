#!/bin/bash
# round 3, sixth GPU pass: deferred evaporation in the interval / run engine
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_f.log 2>&1; tail -3 gpurun_out/pytest_f.log
bash tools/ab_variants.sh smartpy_amd/csrc/libsmart_amd_nodefer.so -- --no-strong 2>&1 | tee gpurun_out/ab_defer.log
for so in default nodefer; do
  if [ "$so" = default ]; then unset SMART_AMD_LIB; else export SMART_AMD_LIB=$PWD/smartpy_amd/csrc/libsmart_amd_$so.so; fi
  echo -n "config 4 $so: "; python bench.py --config 4 --steps 4 --warmup 1 --no-cpu-baseline --no-flat 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.3f ms" % d["roofline"]["launch_ms"], d["roofline"]["kernel"])'
done 2>&1 | tee -a gpurun_out/ab_defer.log
unset SMART_AMD_LIB
python tools/debug/fast_error.py 2>&1 | tail -5 | tee gpurun_out/fast_error_f.log
