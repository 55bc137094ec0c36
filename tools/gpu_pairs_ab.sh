#!/bin/bash
# strides of the pair blocks (library variants from tools/build_variants.py) on the flat-forcing leg, interleaved
export TMPDIR=/tmp
mkdir -p gpurun_out
for rep in 1 2 3; do
  for f in default $(ls tools/variants/libsmart_amd_s*.so); do
    if [ $f = default ]; then unset SMART_AMD_LIB; else export SMART_AMD_LIB=$PWD/$f; fi
    echo -n "$(basename $f .so | sed s/libsmart_amd_//): "; python tools/debug/flat_only.py 100000 8 2>/dev/null | grep " ms" | sort -n | head -3 | tr '\n' ' '; echo
  done
  unset SMART_AMD_LIB
  echo -n "threaded: "; SMART_PAIR_BLOCKS=0 python tools/debug/flat_only.py 100000 8 2>/dev/null | grep " ms" | sort -n | head -3 | tr '\n' ' '; echo
done 2>&1 | tee gpurun_out/pairs_ab.log
