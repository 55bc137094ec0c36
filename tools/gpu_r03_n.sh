#!/bin/bash
# round 3, fourteenth GPU pass: the same crossover with rows ordered (no discharge matrix stored)
export TMPDIR=/tmp
mkdir -p gpurun_out
for rep in 1 2; do for cfg in "--config 4 --samples 100000" "--config 4 --samples 125000" "--config 4 --samples 160000" "--config 4 --samples 200000" "--config 4 --samples 300000"; do for ex in 0 1; do
echo -n "SMART_EXITS=$ex $cfg --no-discharge: "; SMART_EXITS=$ex timeout 300 python bench.py $cfg --no-discharge --steps 4 --warmup 1 --no-cpu-baseline --no-flat --no-strong 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms' % d['roofline']['launch_ms'], d['roofline']['kernel'][:60])"
done; done; done 2>&1 | tee gpurun_out/ab_exits_modes_ordered.log
