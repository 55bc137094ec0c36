#!/bin/bash
# the default slice count (24) against 16 on every leg of the bench line, interleaved
export TMPDIR=/tmp
mkdir -p gpurun_out
for rep in 1 2 3; do for k in default 16; do
  if [ $k = default ]; then unset SMART_TIME_SLICES; else export SMART_TIME_SLICES=$k; fi
  echo -n "slices $k: "; python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-strong 2>/dev/null | tail -1 | python tools/bench_digest.py | grep " ms" | awk '{printf "%s %s | ", $1, $2}'; echo
done; done 2>&1 | tee gpurun_out/slices_ab.log
