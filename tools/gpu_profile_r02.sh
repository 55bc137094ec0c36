#!/bin/bash
# rocprofv3 evidence of round 2: the headline run, one rank's shard of config 4, config 4 on one GPU, the flat leg
export TMPDIR=/tmp
bash tools/profile.sh r02_config3 --steps 20 --warmup 5 --no-cpu-baseline --no-flat
bash tools/profile.sh r02_config4_shard --config 4 --samples 125000 --steps 20 --warmup 5 --no-cpu-baseline --no-flat
bash tools/profile.sh r02_config4_1gpu --config 4 --steps 8 --warmup 3 --no-cpu-baseline --no-flat
bash tools/profile.sh r02_config5_1gpu --config 5 --steps 8 --warmup 3 --no-cpu-baseline --no-flat
bash tools/profile.sh r02_config2 --config 2 --steps 20 --warmup 5 --no-cpu-baseline --no-flat
mkdir -p gpurun_out/prof_r02_flat
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r02_flat/trace -o trace -- python3 tools/debug/flat_only.py 100000 12 > gpurun_out/prof_r02_flat/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/prof_r02_flat/pmc_SQ_WAVES -o pmc -- python3 tools/debug/flat_only.py 100000 12 > gpurun_out/prof_r02_flat/pmc.log 2>&1
du -sh gpurun_out/prof_r02_*
