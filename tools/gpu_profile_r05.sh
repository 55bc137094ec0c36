#!/bin/bash
# rocprofv3 evidence of round 5, ONE gpurun call on the round's final kernels: the headline run, the configurations, the
# step loop at 1e5 and 1e6 samples, the run engine, the raw / every-step kernels; the reciprocal divisions against the
# true ones; the bench line of every configuration.  (The soak is a call of its own: tools/gpu_round.sh soak.)
export TMPDIR=/tmp
mkdir -p gpurun_out
B="--no-cpu-baseline --no-flat --no-strong"
bash tools/profile.sh r05_config3 --steps 20 --warmup 5 $B
bash tools/profile.sh r05_config4_1gpu --config 4 --steps 6 --warmup 2 $B
bash tools/profile.sh r05_config4_shard --config 4 --samples 125000 --steps 12 --warmup 3 $B
bash tools/profile.sh r05_config5_1gpu --config 5 --steps 6 --warmup 2 $B
bash tools/profile.sh r05_config2 --config 2 --steps 20 --warmup 5 $B
bash tools/profile_cmd.sh r05_flat_forcing tools/debug/flat_only.py 100000 12
bash tools/profile_cmd.sh r05_flat_forcing_1e6 tools/debug/flat_only.py 1000000 4
bash tools/profile_cmd.sh r05_runs_of_6 tools/debug/runs_only.py 100000 12
bash tools/profile_cmd.sh r05_raw_gap24 tools/debug/reports_only.py raw 100000 12
bash tools/profile_cmd.sh r05_raw_gap24_flat tools/debug/reports_only.py raw_flat 100000 12
bash tools/profile_cmd.sh r05_gap1 tools/debug/reports_only.py every 100000 8
python tools/debug/recip_bits.py > gpurun_out/r05_recip_bits.txt 2>&1; tail -2 gpurun_out/r05_recip_bits.txt
# round 5: the literal step one sample per DPP row -- the microbenchmark (cycles per step of a lone wavefront, every bit
# against the lane-per-sample form, what the instructions cost), the hook per call, config 2 one class at a time
(cd tools/microbench && { ./lanes 1160 4018 12; echo; ./lanes 1 4018 12; echo; ./lanes 4 4018 12 wet; echo; ./lanes 4 4018 12 dry; echo; ./lanes 4096 4018 96; echo; ./lanes probe; }) > gpurun_out/r05_microbench_lanes.txt 2>&1; tail -3 gpurun_out/r05_microbench_lanes.txt
python tools/debug/hook_time.py > gpurun_out/r05_hook_time.txt 2>&1; tail -4 gpurun_out/r05_hook_time.txt
python tools/debug/illcond_only.py > gpurun_out/r05_config2_classes.txt 2>&1; python tools/debug/illcond_vary.py >> gpurun_out/r05_config2_classes.txt 2>&1; tail -14 gpurun_out/r05_config2_classes.txt
bash tools/gpu_configs.sh r05 > gpurun_out/configs_r05.log 2>&1; tail -30 gpurun_out/configs_r05.log
du -sh gpurun_out
