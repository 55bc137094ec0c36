#!/bin/bash
# bench line of every configuration (1 GPU), default bench, contract test.  usage: bash tools/gpu_configs.sh <tag>
TAG=${1:-b}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_$TAG.log 2>&1; echo "bench rc=$?"
for c in 2 4 5; do timeout 900 python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_${TAG}_c$c.log 2>&1; echo "config $c rc=$?"; done
timeout 900 python bench.py --config 4 --samples 125000 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/bench_${TAG}_c4shard.log 2>&1; echo "config 4 shard rc=$?"
timeout 600 python -m pytest tests/test_bench_contract.py -m gpu -x -q 2>&1 | tail -5
for f in gpurun_out/bench_$TAG*.log; do echo "== $f"; grep '^{' $f | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d['roofline']
    print(d['config']['workload'][:90]); print('  value %.4g  ms/step %.3f  launch_ms %.3f  kernel %s' % (d['value'], d['ms_per_step'], r['launch_ms'], r['kernel']))
    print('  frac', r['frac'], 'alg', r['algorithmic_ratio']['ratio'], 'hbm', r['hbm']['frac'])
    print('  flat', d.get('flat_forcing')); print('  parity', d.get('parity')); print('  cpu', d.get('cpu_baseline'))
"; grep -v '^{' $f | grep -v amdgpu.ids | tail -5; done
