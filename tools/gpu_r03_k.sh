#!/bin/bash
# round 3, eleventh GPU pass: wet intervals in two modes (absorbed by the top layer / full), fill exit in the rain arm.
# Variants (tools/build_variants.py): old = -DSMART_WET_MODES=0 -DSMART_RAIN_FILL_EXIT=0, wm2 = -DSMART_WET_MODES=2,
# fe0 = -DSMART_RAIN_FILL_EXIT=0
export TMPDIR=/tmp
mkdir -p gpurun_out
C=smartpy_amd/csrc
{
  SMART_AMD_LIB=$PWD/$C/libsmart_amd_old.so python tools/debug/steps_bits.py dump /tmp/bits_old.npz
  python tools/debug/steps_bits.py dump /tmp/bits_new.npz
  SMART_AMD_LIB=$PWD/$C/libsmart_amd_wm2.so python tools/debug/steps_bits.py dump /tmp/bits_wm2.npz
  python tools/debug/steps_bits.py compare /tmp/bits_old.npz /tmp/bits_new.npz
  python tools/debug/steps_bits.py compare /tmp/bits_old.npz /tmp/bits_wm2.npz
} 2>&1 | tail -12 | tee gpurun_out/wet_modes_bits.txt
bash tools/ab_variants.sh $C/libsmart_amd_old.so $C/libsmart_amd_wm2.so $C/libsmart_amd_fe0.so -- --no-strong 2>&1 | tee gpurun_out/ab_wet_modes.log
for rep in 1 2; do
  for so in default old; do
    if [ "$so" = default ]; then unset SMART_AMD_LIB; else export SMART_AMD_LIB=$PWD/$C/libsmart_amd_$so.so; fi
    echo -n "$so: flat 1e6 "; python tools/debug/flat_only.py 1000000 3 2>/dev/null | tail -2 | tr '\n' ' '; echo
  done
done 2>&1 | tee -a gpurun_out/ab_wet_modes.log
unset SMART_AMD_LIB
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/wet_modes_tests.txt
