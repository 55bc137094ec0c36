#!/bin/bash
# soak of the ticket-drawn time slices (round 4: with the explicit waits of publish_slice, the raw / every-step kernels,
# and a competitor that holds about half the CUs for the whole launch): every launch with poisoned output buffers,
# compared bit for bit with the unsliced launch, its status word read back
export TMPDIR=/tmp
S=tools/debug/time_slices_stress.py
echo "# $S: time-sliced launches with poisoned output buffers, each compared bit for bit with"
echo "# the unsliced launch and its status word read back (hourly 10 yr + 1 yr warm-up, objectives fused); the hand-over"
echo "# buffers keep their addresses from launch to launch: consumers meet an L1 that is warm with the previous launch's lines"
run() { python $S "$@" 2>/dev/null | tail -1; }
for n in 100000 70000 150000 262144 123457 400000; do run $n 200; done
for n in 100000 150000 262144; do run $n 200 busy; done
for n in 100000 70000 150000 262144; do run $n 200 half; done
# the step loop (forcing that varies inside the day): the pending evaporation demand travels in the hand-over
for n in 100000 150000; do run $n 150 flat; done
run 100000 150 half flat
# the run engine (forcing constant over runs of six steps)
for n in 100000 262144; do run $n 150 runs; done
run 100000 150 half runs
# round 4's kernels: raw reports (interval engine / step loop), a report every step
for n in 100000 150000; do run $n 150 raw; done
run 100000 150 half raw
run 100000 150 flat raw
run 100000 150 half flat raw
run 100000 60 every
run 100000 60 half every
