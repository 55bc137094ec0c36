#!/bin/bash
# rocprofv3 evidence of round 3 (one gpurun call): the headline run, the configurations, the step loop at 1e5 and 1e6
# samples, the run engine; bits of the asm loops against hipcc's; the soak of the time slices; bench lines
export TMPDIR=/tmp
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
# bits: asm step loop against hipcc's loop of round 2; asm wet interval against hipcc's loop; two modes against one
# (variants: tools/build_variants.py oldsteps=-DSMART_STEP_ARMS=0 wetasm0=-DSMART_WET_ASM=0
#  "old=-DSMART_WET_MODES=0 -DSMART_RAIN_FILL_EXIT=0")
( SMART_AMD_LIB=$PWD/smartpy_amd/csrc/libsmart_amd_oldsteps.so timeout 900 python tools/debug/steps_bits.py dump /tmp/b_old.npz
  SMART_AMD_LIB=$PWD/smartpy_amd/csrc/libsmart_amd_wetasm0.so timeout 900 python tools/debug/steps_bits.py dump /tmp/b_wet0.npz
  SMART_AMD_LIB=$PWD/smartpy_amd/csrc/libsmart_amd_old.so timeout 900 python tools/debug/steps_bits.py dump /tmp/b_modes0.npz
  timeout 900 python tools/debug/steps_bits.py dump /tmp/b_new.npz
  echo "== -DSMART_STEP_ARMS=0 (compiled step loop of round 2) against the shipped library"
  python tools/debug/steps_bits.py compare /tmp/b_old.npz /tmp/b_new.npz
  echo "== -DSMART_WET_ASM=0 (hipcc's wet-interval loop) against the shipped library"
  python tools/debug/steps_bits.py compare /tmp/b_wet0.npz /tmp/b_new.npz
  echo "== -DSMART_WET_MODES=0 -DSMART_RAIN_FILL_EXIT=0 (every wet step fills all six layers) against the shipped library"
  python tools/debug/steps_bits.py compare /tmp/b_modes0.npz /tmp/b_new.npz ) > gpurun_out/r03_steps_bits.txt 2>&1
tail -9 gpurun_out/r03_steps_bits.txt
python tools/debug/recip_bits.py > gpurun_out/r03_recip_bits.txt 2>&1; tail -2 gpurun_out/r03_recip_bits.txt
bash tools/gpu_configs.sh r03 > gpurun_out/configs_r03.log 2>&1; tail -30 gpurun_out/configs_r03.log
timeout 1500 bash tools/gpu_soak.sh > gpurun_out/r03_time_slice_soak.txt 2>&1; tail -5 gpurun_out/r03_time_slice_soak.txt
du -sh gpurun_out
