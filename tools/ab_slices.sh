#!/bin/bash
# Slice counts of the time-sliced launch, interleaved on one box (SMART_TIME_SLICES is read when the ABI field is 0).
# usage (on the GPU box): bash tools/ab_slices.sh [bench args]
for rep in 1 2 3; do
for cfg in "default:" "K8:SMART_TIME_SLICES=8" "K24:SMART_TIME_SLICES=24" "K32:SMART_TIME_SLICES=32" "K48:SMART_TIME_SLICES=48" "off:SMART_TIME_SLICES=0"; do
name=${cfg%%:*}; envs=${cfg#*:}
echo -n "$name: "; env $envs timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-flat "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms  %s' % (d['roofline']['launch_ms'], d['roofline']['kernel']))"
done; done
