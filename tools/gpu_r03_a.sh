#!/bin/bash
# round 3, first GPU pass: asm arms of the step loop against the compiled loop (bits), the GPU suite, A/B timing
export TMPDIR=/tmp
mkdir -p gpurun_out
( SMART_AMD_LIB=$PWD/smartpy_amd/csrc/libsmart_amd_oldsteps.so timeout 900 python tools/debug/steps_bits.py dump /tmp/steps_old.npz
  timeout 900 python tools/debug/steps_bits.py dump /tmp/steps_new.npz
  python tools/debug/steps_bits.py compare /tmp/steps_old.npz /tmp/steps_new.npz ) > gpurun_out/steps_bits.log 2>&1
tail -5 gpurun_out/steps_bits.log
bash tools/ab_variants.sh smartpy_amd/csrc/libsmart_amd_oldsteps.so > gpurun_out/ab_arms.log 2>&1; cat gpurun_out/ab_arms.log
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_a.log 2>&1; tail -8 gpurun_out/pytest_a.log
