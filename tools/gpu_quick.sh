#!/bin/bash
# quick A/B of one build: the GPU test-suite, then launch time of the headline run and of the flat-forcing leg
export TMPDIR=/tmp
TAG=${1:-q}
timeout 2000 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_$TAG.log 2>&1; tail -5 gpurun_out/pytest_$TAG.log
python bench.py --steps 6 --warmup 1 --no-cpu-baseline ${@:2} > gpurun_out/bench_$TAG.log 2>&1
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/bench_$TAG.log") if l.startswith("{")][0])
print("headline launch_ms", d["roofline"]["launch_ms"], d["roofline"]["kernel"])
f=d.get("flat_forcing"); print("flat launch_ms", f and f["launch_ms"], f and f["kernel"])
PY
