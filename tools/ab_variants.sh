#!/bin/bash
# A/B kernel-tuning variants of libsmart_amd.so on the bench workload, interleaved, in one process sequence.
# usage (on the GPU box): bash tools/ab_variants.sh <variant.so> [<variant.so> ...]   (built by tools/build_variants.py)
for rep in 1 2 3; do
  for so in default "$@"; do
    if [ "$so" = default ]; then unset SMART_AMD_LIB; else export SMART_AMD_LIB=$PWD/$so; fi
    echo -n "$so: "; timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms  %.4g steps/s' % (d['roofline']['launch_ms'], d['value']))"
  done
done
