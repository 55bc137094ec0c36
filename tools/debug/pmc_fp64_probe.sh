export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc64
mkdir -p $OUT
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT64 --kernel-trace --output-format csv -d $OUT -o pmc -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-flat --no-strong > $OUT/log.txt 2>&1
echo rc=$?
python3 - <<'PY'
import csv, glob, collections
rows = []
for p in glob.glob('gpurun_out/pmc64/**/*counter_collection.csv', recursive=True):
    rows += list(csv.DictReader(open(p)))
acc = collections.defaultdict(list)
for r in rows:
    if 'smart_fast' in r['Kernel_Name'] and int(r['Grid_Size']) > 100000:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(acc.items()): print(k, len(v), sum(v)/len(v))
PY
