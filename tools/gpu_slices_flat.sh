#!/bin/bash
# slice counts on the flat-forcing leg (the pair blocks changed what a slice costs) and on the headline
export TMPDIR=/tmp
mkdir -p gpurun_out
for rep in 1 2; do for k in 8 12 16 20 24 32 48; do
  echo -n "flat slices $k: "; SMART_TIME_SLICES=$k python tools/debug/flat_only.py 100000 6 2>/dev/null | grep " ms" | sort -n | head -3 | tr '\n' ' '; echo
done; done 2>&1 | tee gpurun_out/slices_flat.log
