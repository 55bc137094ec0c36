#!/bin/bash
# slice counts on the headline run (interval engine) and the run engine
export TMPDIR=/tmp
mkdir -p gpurun_out
for rep in 1 2; do for k in 12 16 20 24 32; do
  echo -n "headline slices $k: "; SMART_TIME_SLICES=$k python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-flat --no-strong 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms' % d['roofline']['launch_ms'])"
  echo -n "runs6 slices $k: "; SMART_TIME_SLICES=$k python tools/debug/runs_only.py 100000 6 2>/dev/null | grep " ms" | sort -n | head -3 | tr '\n' ' '; echo
done; done 2>&1 | tee gpurun_out/slices_headline.log
