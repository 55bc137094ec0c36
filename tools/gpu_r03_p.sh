#!/bin/bash
# round 3, sixteenth GPU pass: the observation of the next report interval requested a report ahead in the step loops
# (run_ensemble, step loop of the merged kernels, run engine).  base = the tree before (tools/build_variants.py)
export TMPDIR=/tmp
mkdir -p gpurun_out
C=smartpy_amd/csrc
bash tools/ab_variants.sh $C/libsmart_amd_base.so -- --no-strong 2>&1 | tee gpurun_out/ab_obs_ahead.log
for rep in 1 2 3; do for so in default base; do
  if [ "$so" = default ]; then unset SMART_AMD_LIB; else export SMART_AMD_LIB=$PWD/$C/libsmart_amd_$so.so; fi
  echo -n "$so: config 2 "; python bench.py --config 2 --steps 20 --warmup 5 --no-cpu-baseline --no-flat --no-strong 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms/step  launch %.3f' % (d['ms_per_step'], d['roofline']['launch_ms']))"
  echo -n "$so: flat 1e6 "; python tools/debug/flat_only.py 1000000 3 2>/dev/null | tail -2 | tr '\n' ' '; echo
done; done 2>&1 | tee -a gpurun_out/ab_obs_ahead.log
unset SMART_AMD_LIB
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
