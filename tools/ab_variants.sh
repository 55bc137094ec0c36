#!/bin/bash
# A/B builds of libsmart_amd.so on the bench workload, interleaved, on one box.
# usage (on the GPU box): bash tools/ab_variants.sh <variant.so> [<variant.so> ...] [-- bench args]
LIBS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done; [ "$1" = "--" ] && shift
for rep in 1 2 3; do
  for so in default "${LIBS[@]}"; do
    if [ "$so" = default ]; then unset SMART_AMD_LIB; else export SMART_AMD_LIB=$PWD/$so; fi
    echo -n "$so: "; timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c '
import sys, json
d = json.loads(sys.stdin.read()); f = d.get("flat_forcing"); r = d.get("runs_of_6"); o = d.get("objectives_only")
ms = lambda x: "%.3f" % x["launch_ms"] if x else "-"
print("%.3f ms  %.4g steps/s   flat %s ms   runs6 %s ms   objectives only %s ms" % (d["roofline"]["launch_ms"], d["value"], ms(f), ms(r), ms(o)))'
  done
done
