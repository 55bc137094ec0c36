#!/bin/bash
# The N > 1 code path of bench.py on a ONE-GPU box: two ranks share GPU 0, the collectives go through gloo (host
# staged) instead of RCCL -- same sharding, gathering, timing and JSON code as the driver's multi-GPU run.
export TMPDIR=/tmp SMART_DIST_BACKEND=gloo
for args in "--config 3 --samples 30000" "--config 4 --samples 60000" "--config 5 --samples 2000" "--config 2"; do
  echo "== $args"
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $(python -c "import socket; s = socket.socket(); s.bind(('127.0.0.1', 0)); print(s.getsockname()[1])") \
     bench.py --gpus 2 --steps 3 --warmup 1 $args 2>&1 | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(d['config']['workload'][:110]); print('  n_gpus', d['n_gpus'], 'scaling', d['scaling'], 'value %.4g' % d['value'], 'ms/step %.3f' % d['ms_per_step'], d['config']['parallelism'], 'runs_total', d['config']['runs_total'], 'per gpu', d['config']['runs_per_gpu'])"
done
