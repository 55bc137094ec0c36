export TMPDIR=/tmp
for rep in 1 2; do for ex in 0 1; do for cfg in "--config 3" "--config 4 --samples 125000" "--config 4 --samples 160000" "--config 4 --samples 200000"; do
echo -n "SMART_EXITS=$ex $cfg: "; SMART_EXITS=$ex python bench.py $cfg --steps 5 --warmup 1 --no-cpu-baseline --no-flat 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms' % d['roofline']['launch_ms'], d['roofline']['kernel'][:40])"
done; done; done
