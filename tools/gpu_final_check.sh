#!/bin/bash
# what the driver runs at round end, on the tree as committed: build check, smoke, GPU suite, default bench; plus the
# two-rank (gloo, one GPU) path of bench.py
export TMPDIR=/tmp
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_final.log 2>&1; tail -3 gpurun_out/pytest_final.log
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_final.log 2>&1; grep '^{' gpurun_out/bench_final.log | python -c '
import sys, json
d = json.loads(sys.stdin.read()); r = d["roofline"]
print("value %.4g  ms/step %.3f  frac %s useful %s held %s traffic %s" % (d["value"], d["ms_per_step"], r["frac"], r["useful_frac"], r["frac_at_held_clock"], r["traffic"]))
print("pmc:", r["pmc_source"][:100])
for k in ("flat_forcing", "runs_of_6", "objectives_only"): print(k, "%.3f ms" % d[k]["launch_ms"], d[k]["kernel"])
print("strong_1e6 %.3f ms" % d["strong_1e6"]["ms_per_step"], d["strong_1e6"]["kernel"]); print("parity", d["parity"]["max_rel_discharge"], "cpu %.4g" % d["cpu_baseline"]["value"])'
bash tools/gpu_two_ranks_one_gpu.sh 2>&1 | tail -12
