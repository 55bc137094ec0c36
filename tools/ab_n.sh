for rep in 1 2; do for so in "" smartpy_amd/csrc/libsmart_amd_w4.so; do for n in 100000 1000000; do
echo -n "${so:-default} N=$n: "; SMART_AMD_LIB=${so:+$PWD/$so} python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-discharge --samples $n 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms' % d['roofline']['launch_ms'])"; done; done; done
