#!/bin/bash
# rocprofv3 evidence of round 6, ONE gpurun call on the round's final kernels: the headline run, the configurations and
# the legs the bench line prices (their PMC counts are hash-guarded: profiles/traffic_latest.json) -- and, new this
# round, DAILY ensembles of 1e4 / 1e5 / 1e6 samples with the literal rows in both forms (per-class kernel times from the
# kernel trace; the counters for the form the library picks).
export TMPDIR=/tmp
mkdir -p gpurun_out
B="--no-cpu-baseline --no-flat --no-strong --no-daily"
bash tools/profile.sh r06_config3 --steps 20 --warmup 5 $B
bash tools/profile.sh r06_config4_1gpu --config 4 --steps 6 --warmup 2 $B
bash tools/profile.sh r06_config4_shard --config 4 --samples 125000 --steps 12 --warmup 3 $B
bash tools/profile.sh r06_config5_1gpu --config 5 --steps 6 --warmup 2 $B
bash tools/profile.sh r06_config2 --config 2 --steps 20 --warmup 5 $B
bash tools/profile_cmd.sh r06_flat_forcing tools/debug/flat_only.py 100000 12
bash tools/profile_cmd.sh r06_runs_of_6 tools/debug/runs_only.py 100000 12
bash tools/profile_cmd.sh r06_raw_gap24 tools/debug/reports_only.py raw 100000 12
bash tools/profile_cmd.sh r06_gap1 tools/debug/reports_only.py every 100000 8
# daily ensembles: the form the library picks with every counter pass; the other form's kernel trace beside it
bash tools/profile_cmd.sh r06_daily_1e4 tools/debug/daily_only.py 10000 12 auto
TRACE_ONLY=1 bash tools/profile_cmd.sh r06_daily_1e4_lanes tools/debug/daily_only.py 10000 12 lanes
bash tools/profile_cmd.sh r06_daily_1e5 tools/debug/daily_only.py 100000 12 auto
TRACE_ONLY=1 bash tools/profile_cmd.sh r06_daily_1e5_rows tools/debug/daily_only.py 100000 12 rows
bash tools/profile_cmd.sh r06_daily_1e6 tools/debug/daily_only.py 1000000 8 auto
TRACE_ONLY=1 bash tools/profile_cmd.sh r06_daily_1e6_rows tools/debug/daily_only.py 1000000 4 rows
python tools/debug/daily_forms.py --classes > gpurun_out/r06_daily_form.txt 2>&1; tail -12 gpurun_out/r06_daily_form.txt
python tools/debug/hook_time.py > gpurun_out/r06_hook_time.txt 2>&1; tail -4 gpurun_out/r06_hook_time.txt
bash tools/gpu_configs.sh r06 > gpurun_out/configs_r06.log 2>&1; tail -30 gpurun_out/configs_r06.log
du -sh gpurun_out
