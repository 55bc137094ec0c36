for rep in 1 2; do for so in "" smartpy_amd/csrc/libsmart_amd_w4.so; do
echo "== ${so:-default}"; SMART_AMD_LIB=${so:+$PWD/$so} timeout 300 python tools/debug/time_slices_sweep.py 100000 262144 1000000 2>&1 | grep -v amdgpu.ids; done; done
