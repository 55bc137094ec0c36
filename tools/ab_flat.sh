for ts in default 0; do
echo "== SMART_TIME_SLICES=$ts"; if [ $ts = default ]; then unset SMART_TIME_SLICES; else export SMART_TIME_SLICES=$ts; fi
timeout 300 python tools/debug/flat_path_cost.py 2>&1 | grep -v amdgpu.ids; done
