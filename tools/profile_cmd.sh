#!/bin/bash
# rocprofv3 evidence for any python script: kernel-trace stats, then the PMC passes of tools/profile.sh, each in its own run.
# usage: bash tools/profile_cmd.sh <tag> <script.py> [args...]     (the program itself stands behind `--`: no shell hop)
# TRACE_ONLY=1: the kernel trace alone (per-kernel times of an A/B's other side); ONLY=<first counter>: that pass alone
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
python3 -c "import bench; print(bench.kernel_source_hash())" > $OUT/source_hash.txt
echo "$@" > $OUT/args.txt
if [ -z "$ONLY" ]; then rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 "$@" > $OUT/trace.log 2>&1
echo "trace rc=$?"; fi
[ -n "$TRACE_ONLY" ] && { du -sh $OUT; exit 0; }
for PASS in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_BRANCH" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT64"; do
  NAME=$(echo $PASS | cut -d' ' -f1)
  [ -n "$ONLY" ] && [ "$ONLY" != "$NAME" ] && continue
  rocprofv3 --pmc $PASS --kernel-trace --output-format csv -d $OUT/pmc_$NAME -o pmc -- python3 "$@" > $OUT/pmc_${NAME}.log 2>&1
  echo "pmc $NAME rc=$?"
done
du -sh $OUT
