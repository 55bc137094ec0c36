for rep in 1 2 3; do
for cfg in "default:" "K24:SMART_TIME_SLICES=24" "K32:SMART_TIME_SLICES=32" "K48:SMART_TIME_SLICES=48" "sleep20:SMART_AMD_LIB=$PWD/smartpy_amd/csrc/libsmart_amd_sl20.so"; do
name=${cfg%%:*}; envs=${cfg#*:}
echo -n "$name: "; env $envs timeout 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms' % d['roofline']['launch_ms'])"
done; done
