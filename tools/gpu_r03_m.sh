#!/bin/bash
# round 3, thirteenth GPU pass: the kernel with exits with the asm wet loop (SMART_WET_ASM=2), as hipcc allocates it
# (193 VGPRs, two waves: wa2) and bound to three waves per SIMD (167 VGPRs, 84 spilled to scratch: wa3)
export TMPDIR=/tmp
mkdir -p gpurun_out
C=smartpy_amd/csrc
for rep in 1 2; do for cfg in "--config 4 --samples 160000" "--config 4 --samples 200000" "--config 4 --samples 300000" "--config 4 --samples 1000000" "--config 5"; do for so in default wa2 wa3; do
if [ "$so" = default ]; then unset SMART_AMD_LIB; else export SMART_AMD_LIB=$PWD/$C/libsmart_amd_$so.so; fi
echo -n "$so SMART_EXITS=1 $cfg: "; SMART_EXITS=1 timeout 300 python bench.py $cfg --steps 4 --warmup 1 --no-cpu-baseline --no-flat --no-strong 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms' % d['roofline']['launch_ms'], d['roofline']['kernel'][:60], d['parity'] if 'parity' in d else '')"
done; done; done 2>&1 | tee gpurun_out/ab_exits_asm.log
