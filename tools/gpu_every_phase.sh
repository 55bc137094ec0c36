#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
for rep in 1 2; do for f in default $(ls tools/variants/libsmart_amd_e*.so); do
  if [ $f = default ]; then unset SMART_AMD_LIB; else export SMART_AMD_LIB=$PWD/$f; fi
  echo -n "$(basename $f .so | sed s/libsmart_amd_//) every: "; python tools/debug/reports_only.py every 100000 6 2>/dev/null | grep " ms" | sort -n | head -3 | tr '\n' ' '; echo
done; done 2>&1 | tee gpurun_out/every_phase.log
unset SMART_AMD_LIB
bash tools/gpu_slices_flat.sh
