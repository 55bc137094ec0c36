#!/bin/bash
# round 3, fifteenth GPU pass: the fp64 instruction-class counters (SQ_INSTS_VALU_{FMA,ADD,MUL,TRANS}_F64) of every
# workload of tools/gpu_profile_r03.sh, added to its profiles (same kernels: the source hash is checked by the summary)
export TMPDIR=/tmp ONLY=SQ_INSTS_VALU_FMA_F64
mkdir -p gpurun_out
B="--no-cpu-baseline --no-flat --no-strong"
bash tools/profile.sh r03_config3 --steps 20 --warmup 5 $B
bash tools/profile.sh r03_config4_1gpu --config 4 --steps 6 --warmup 2 $B
bash tools/profile.sh r03_config4_shard --config 4 --samples 125000 --steps 12 --warmup 3 $B
bash tools/profile.sh r03_config5_1gpu --config 5 --steps 6 --warmup 2 $B
bash tools/profile.sh r03_config2 --config 2 --steps 20 --warmup 5 $B
bash tools/profile_cmd.sh r03_flat_forcing tools/debug/flat_only.py 100000 12
bash tools/profile_cmd.sh r03_flat_forcing_1e6 tools/debug/flat_only.py 1000000 4
bash tools/profile_cmd.sh r03_runs_of_6 tools/debug/runs_only.py 100000 12
