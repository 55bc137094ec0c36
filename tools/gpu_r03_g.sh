#!/bin/bash
# round 3, seventh GPU pass: rainless runs classified on the scalar unit, calm intervals without filling
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_g.log 2>&1; tail -3 gpurun_out/pytest_g.log
bash tools/ab_variants.sh smartpy_amd/csrc/libsmart_amd_prev.so -- --no-strong 2>&1 | tee gpurun_out/ab_calm.log
