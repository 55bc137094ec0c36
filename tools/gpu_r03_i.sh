#!/bin/bash
# round 3, ninth GPU pass: unguarded leaks / max-clamps of the reciprocal path; build + smoke in one process
export TMPDIR=/tmp
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_i.log 2>&1; tail -3 gpurun_out/pytest_i.log
for rep in 1 2 3; do
  for so in default final2; do
    if [ "$so" = default ]; then unset SMART_AMD_LIB; else export SMART_AMD_LIB=$PWD/smartpy_amd/csrc/libsmart_amd_$so.so; fi
    echo -n "config 2 $so: "; python bench.py --config 2 --steps 10 --warmup 2 --no-cpu-baseline --no-flat 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.3f ms" % d["roofline"]["launch_ms"])'
  done
done 2>&1 | tee gpurun_out/ab_unguarded.log
unset SMART_AMD_LIB
python tools/debug/recip_bits.py 2>&1 | tail -3 | tee gpurun_out/r03_recip_bits.txt
