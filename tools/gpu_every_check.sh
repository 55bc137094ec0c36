#!/bin/bash
# the every-step stream (SMART_A_EVERY_STREAM) against the step-by-step loop of the same library (SMART_PAIR_BLOCKS=0)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q -k "every or pair_blocks or raw_and or bench_legs" 2>&1 | tail -4
python tools/debug/steps_bits.py dump /tmp/bits_new.npz > /dev/null || echo "dump failed"
SMART_PAIR_BLOCKS=0 python tools/debug/steps_bits.py dump /tmp/bits_old.npz > /dev/null || echo "dump (old) failed"
python tools/debug/steps_bits.py compare /tmp/bits_new.npz /tmp/bits_old.npz 2>&1 | tail -3
for rep in 1 2; do
  echo -n "stream every: "; python tools/debug/reports_only.py every 100000 6 2>/dev/null | grep " ms" | sort -n | head -3 | tr '\n' ' '; echo
  echo -n "step-by-step every: "; SMART_PAIR_BLOCKS=0 python tools/debug/reports_only.py every 100000 6 2>/dev/null | grep " ms" | sort -n | head -3 | tr '\n' ' '; echo
  echo -n "pairs raw_flat: "; python tools/debug/reports_only.py raw_flat 100000 6 2>/dev/null | grep " ms" | sort -n | head -3 | tr '\n' ' '; echo
done
