#!/bin/bash
# condense what tools/gpu_profile_r06.sh brought back into profiles/ (run in the build container, after the gpurun call)
S=tools/summarize_profile.py
# the instruction models first: the summaries quote their fp64 shares (needs profiles/r03_fill_paths.json: tools/fill_paths.py)
python tools/isa_model.py intervals profiles/r06_isa_model_intervals > /dev/null 2>&1
python tools/isa_model.py steps profiles/r06_isa_model_steps > /dev/null 2>&1
python $S gpurun_out/prof_r06_config3 profiles/r06_config3 "config3:runs_per_gpu=100000:discharge=1:math=fast" > /dev/null
python $S gpurun_out/prof_r06_config4_shard profiles/r06_config4_shard "config4:runs_per_gpu=125000:discharge=0:math=fast" > /dev/null
python $S gpurun_out/prof_r06_config4_1gpu profiles/r06_config4_1gpu "config4:runs_per_gpu=1000000:discharge=0:math=fast" > /dev/null
python $S gpurun_out/prof_r06_config5_1gpu profiles/r06_config5_1gpu "config5:runs_per_gpu=640000:discharge=0:math=fast" > /dev/null
python $S gpurun_out/prof_r06_config2 profiles/r06_config2 "config2:runs_per_gpu=10000:discharge=1:math=fast" > /dev/null
# (the legs of the bench line under keys of their own in profiles/traffic_latest.json: bench.py prices each leg's launch
# against the issue roof with its own instruction count)
for t in flat_forcing runs_of_6 raw_gap24 gap1; do python $S gpurun_out/prof_r06_$t profiles/r06_$t "leg:$t" > /dev/null; done
# daily ensembles: the form the library picks (counters), the other form's kernel trace beside it
python $S gpurun_out/prof_r06_daily_1e4 profiles/r06_daily_1e4 > /dev/null
python $S gpurun_out/prof_r06_daily_1e5 profiles/r06_daily_1e5 > /dev/null
python $S gpurun_out/prof_r06_daily_1e6 profiles/r06_daily_1e6 > /dev/null
for t in daily_1e4_lanes daily_1e5_rows daily_1e6_rows; do
  python $S gpurun_out/prof_r06_$t /tmp/r06_$t > /dev/null
  base=${t%_*}; { echo; echo "## the same ensemble with the literal rows in the OTHER form (\`${t##*_}\`): kernel trace only"; echo
    sed -n '/kernel stats/,$p' /tmp/r06_$t.md | sed -n '3,8p'; } >> profiles/r06_$base.md
done
# a daily launch is three kernels side by side: their instructions summed against the issue roof (and the bench leg's key)
python tools/daily_roofline.py gpurun_out/prof_r06_daily_1e4 profiles/r06_daily_1e4 > /dev/null
python tools/daily_roofline.py gpurun_out/prof_r06_daily_1e5 profiles/r06_daily_1e5 > /dev/null
python tools/daily_roofline.py gpurun_out/prof_r06_daily_1e6 profiles/r06_daily_1e6 "leg:daily_1e6" > /dev/null
for f in r06_daily_form r06_hook_time r06_time_slice_soak; do [ -f gpurun_out/$f.txt ] && cp gpurun_out/$f.txt profiles/; done
python tools/kernel_hashes.py > profiles/r06_kernel_hashes.txt 2>&1
for c in "" _c2 _c4 _c5 _c4shard; do grep '^{' gpurun_out/bench_r06$c.log > profiles/r06_bench_${c#_}.jsonl; done
mv profiles/r06_bench_.jsonl profiles/r06_bench_config3.jsonl
# block tables of the hot kernels (the listings themselves stay out of the tree: python tools/isa_report.py <kernel> /tmp/x --hot)
for k in intervals steps; do python tools/isa_report.py smart_fast_$k profiles/r06_isa_$k --hot > /dev/null; rm -f profiles/r06_isa_$k.s; done
python tools/kernel_resources.py > profiles/r06_kernel_resources.txt 2>/dev/null
for t in config3 flat_forcing runs_of_6 raw_gap24 gap1 config2 config4_1gpu config4_shard config5_1gpu daily_1e4 daily_1e5 daily_1e6; do
  echo "== $t"; grep -A4 "clock held and issue" profiles/r06_$t.md | tail -2; grep "timed steps\|traffic (corrected)" profiles/r06_$t.md; done
cat gpurun_out/prof_r06_config3/source_hash.txt; python -c "import bench; print(bench.kernel_source_hash())"
