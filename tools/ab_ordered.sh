#!/bin/bash
# A/B builds of the library on ORDERED rows without stored discharge (tools/debug/sort_rows.py: the engine's own
# ordering is off there, the tool orders by T bins then S*Z).  usage: bash tools/ab_ordered.sh "<sizes>" <name> ...
export TMPDIR=/tmp
SIZES=$1; shift
for rep in 1 2; do for so in default "$@"; do
if [ $so = default ]; then unset SMART_AMD_LIB; else export SMART_AMD_LIB=$PWD/smartpy_amd/csrc/libsmart_amd_$so.so; fi
echo "== $so"; python tools/debug/sort_rows.py $SIZES 2>&1 | grep "T in 64 bins, then S\*Z " | awk 'NR%2==0'
done; done
