#!/bin/bash
# soak of the ticket-drawn time slices: 3,000 launches alone, 1,500 beside a competing stream, 1,200 of the step loop,
# 800 of the run engine
export TMPDIR=/tmp
echo "# tools/debug/time_slices_stress.py: time-sliced launches with poisoned output buffers, each compared bit for bit with"
echo "# the unsliced launch and its status word read back (hourly 10 yr + 1 yr warm-up, objectives fused)"
for n in 100000 70000 150000 262144 66000 123457 400000 90001 131073 200000; do python tools/debug/time_slices_stress.py $n 300 2>/dev/null | tail -1; done
for n in 100000 70000 150000 123457 262144; do python tools/debug/time_slices_stress.py $n 300 busy 2>/dev/null | tail -1; done
# the step loop (forcing that varies inside the day): the pending evaporation demand travels in the hand-over
for n in 100000 70000 150000 262144; do python tools/debug/time_slices_stress.py $n 200 flat 2>/dev/null | tail -1; done
for n in 100000 150000; do python tools/debug/time_slices_stress.py $n 200 busy flat 2>/dev/null | tail -1; done
# the run engine (forcing constant over runs of six steps)
for n in 100000 150000 262144; do python tools/debug/time_slices_stress.py $n 200 runs 2>/dev/null | tail -1; done
python tools/debug/time_slices_stress.py 100000 200 busy runs 2>/dev/null | tail -1
