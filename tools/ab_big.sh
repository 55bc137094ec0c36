# usage: bash tools/ab_big.sh <variant.so> ...   -- plain / sliced launch times at 1e5, 262144 and 1e6 samples
for rep in 1 2; do for so in "" "$@"; do
echo "== ${so:-default}"; SMART_AMD_LIB=${so:+$PWD/$so} timeout 300 python tools/debug/time_slices_sweep.py 100000 262144 1000000 2>&1 | grep -v amdgpu.ids; done; done
