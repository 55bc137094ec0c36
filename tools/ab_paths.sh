for rep in 1 2; do for so in "" smartpy_amd/csrc/libsmart_amd_w3.so; do
echo "== ${so:-default}"; SMART_AMD_LIB=${so:+$PWD/$so} timeout 300 python tools/debug/flat_path_cost.py 2>&1 | grep -v amdgpu.ids | grep "False"; done; done
