#!/bin/bash
# round 3, tenth GPU pass: rain arm without zeros / EXEC region when every lane is wet
export TMPDIR=/tmp
mkdir -p gpurun_out
for rep in 1 2 3; do
  for so in default base; do
    if [ "$so" = default ]; then unset SMART_AMD_LIB; else export SMART_AMD_LIB=$PWD/smartpy_amd/csrc/libsmart_amd_$so.so; fi
    echo -n "$so: 1e5 "; python tools/debug/flat_only.py 100000 4 2>/dev/null | tail -2 | tr '\n' ' '; echo -n " 1e6 "; python tools/debug/flat_only.py 1000000 3 2>/dev/null | tail -2 | tr '\n' ' '; echo
  done
done 2>&1 | tee gpurun_out/ab_allwet.log
unset SMART_AMD_LIB
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "asm_loops or randomized or capacity" 2>&1 | tail -2
