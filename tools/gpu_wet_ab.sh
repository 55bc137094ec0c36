#!/bin/bash
# the wet interval's loops: one step per turn (wm1), four (default), eight (t8); headline / objectives only / raw, interleaved
export TMPDIR=/tmp
mkdir -p gpurun_out
for rep in 1 2 3 4; do for f in default $(ls tools/variants/libsmart_amd_*.so); do
  if [ $f = default ]; then unset SMART_AMD_LIB; else export SMART_AMD_LIB=$PWD/$f; fi
  echo -n "$(basename $f .so | sed s/libsmart_amd_//): "; python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-strong 2>/dev/null | tail -1 | python tools/bench_digest.py | grep " ms" | awk '{printf "%s %s | ", $1, $2}'; echo
done; done 2>&1 | tee gpurun_out/wet_ab.log
