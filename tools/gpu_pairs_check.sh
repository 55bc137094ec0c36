#!/bin/bash
# the pair blocks of the streaming step loop against the threaded chunks, one library (SMART_PAIR_BLOCKS=0 turns them
# off at run time): every output of the merged kernels bit for bit, then the flat-forcing and raw-flat legs both ways
export TMPDIR=/tmp
mkdir -p gpurun_out
python tools/debug/steps_bits.py dump /tmp/bits_new.npz > /dev/null || echo "dump (pairs) failed"
SMART_PAIR_BLOCKS=0 python tools/debug/steps_bits.py dump /tmp/bits_old.npz > /dev/null || echo "dump (threaded) failed"
python tools/debug/steps_bits.py compare /tmp/bits_new.npz /tmp/bits_old.npz 2>&1 | tail -4
for rep in 1 2; do
  echo -n "pairs    flat: "; python tools/debug/flat_only.py 100000 8 2>/dev/null | grep " ms" | sort -n | head -3 | tr '\n' ' '; echo
  echo -n "threaded flat: "; SMART_PAIR_BLOCKS=0 python tools/debug/flat_only.py 100000 8 2>/dev/null | grep " ms" | sort -n | head -3 | tr '\n' ' '; echo
done
echo -n "pairs    raw_flat: "; python tools/debug/reports_only.py raw_flat 2>/dev/null | grep " ms" | sort -n | head -3 | tr '\n' ' '; echo
echo -n "threaded raw_flat: "; SMART_PAIR_BLOCKS=0 python tools/debug/reports_only.py raw_flat 2>/dev/null | grep " ms" | sort -n | head -3 | tr '\n' ' '; echo
echo -n "pairs    flat 1e6: "; python tools/debug/flat_only.py 1000000 4 2>/dev/null | grep " ms" | sort -n | head -3 | tr '\n' ' '; echo
echo -n "threaded flat 1e6: "; SMART_PAIR_BLOCKS=0 python tools/debug/flat_only.py 1000000 4 2>/dev/null | grep " ms" | sort -n | head -3 | tr '\n' ' '; echo
