#!/bin/bash
# round-2 GPU pass: parity tests, bench line, the other configurations.  usage: bash tools/gpu_r02.sh <tag>
TAG=${1:-a}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest_$TAG.log
timeout 600 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_$TAG.log 2>&1; echo "bench rc=$?" | tee -a gpurun_out/bench_$TAG.log
timeout 900 python tools/bench_configs.py > gpurun_out/configs_$TAG.log 2>&1; echo "configs rc=$?" | tee -a gpurun_out/configs_$TAG.log
tail -15 gpurun_out/pytest_$TAG.log; tail -3 gpurun_out/bench_$TAG.log; tail -6 gpurun_out/configs_$TAG.log
