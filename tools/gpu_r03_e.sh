#!/bin/bash
# round 3, fifth GPU pass: the wet interval of the interval engine as an asm loop, against hipcc's loop
export TMPDIR=/tmp
mkdir -p gpurun_out
( SMART_AMD_LIB=$PWD/smartpy_amd/csrc/libsmart_amd_wetasm0.so timeout 900 python tools/debug/steps_bits.py dump /tmp/bits_wetasm0.npz
  timeout 900 python tools/debug/steps_bits.py dump /tmp/bits_new.npz
  python tools/debug/steps_bits.py compare /tmp/bits_wetasm0.npz /tmp/bits_new.npz ) > gpurun_out/bits_wetasm.log 2>&1
tail -2 gpurun_out/bits_wetasm.log
bash tools/ab_variants.sh smartpy_amd/csrc/libsmart_amd_wetasm0.so -- --no-strong 2>&1 | tee gpurun_out/ab_wetasm.log
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_e.log 2>&1; tail -3 gpurun_out/pytest_e.log
