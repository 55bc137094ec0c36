#!/bin/bash
# condense what tools/gpu_profile_r05.sh brought back into profiles/ (run in the build container, after the gpurun call)
S=tools/summarize_profile.py
# the instruction models first: the summaries quote their fp64 shares (needs profiles/r03_fill_paths.json: tools/fill_paths.py)
python tools/isa_model.py intervals profiles/r05_isa_model_intervals > /dev/null 2>&1
python tools/isa_model.py steps profiles/r05_isa_model_steps > /dev/null 2>&1
python $S gpurun_out/prof_r05_config3 profiles/r05_config3 "config3:runs_per_gpu=100000:discharge=1:math=fast" > /dev/null
python $S gpurun_out/prof_r05_config4_shard profiles/r05_config4_shard "config4:runs_per_gpu=125000:discharge=0:math=fast" > /dev/null
python $S gpurun_out/prof_r05_config4_1gpu profiles/r05_config4_1gpu "config4:runs_per_gpu=1000000:discharge=0:math=fast" > /dev/null
python $S gpurun_out/prof_r05_config5_1gpu profiles/r05_config5_1gpu "config5:runs_per_gpu=640000:discharge=0:math=fast" > /dev/null
python $S gpurun_out/prof_r05_config2 profiles/r05_config2 "config2:runs_per_gpu=10000:discharge=1:math=fast" > /dev/null
# (the legs of the bench line under keys of their own in profiles/traffic_latest.json: bench.py prices each leg's launch
# against the issue roof with its own instruction count)
for t in flat_forcing runs_of_6 raw_gap24 gap1; do python $S gpurun_out/prof_r05_$t profiles/r05_$t "leg:$t" > /dev/null; done
for t in flat_forcing_1e6 raw_gap24_flat; do python $S gpurun_out/prof_r05_$t profiles/r05_$t > /dev/null; done
# the three kernels that took over from smart_fast_plain, side by side (the verdict's profiles/r05_plain.md)
{ echo "# Raw reports and a report every step (round 5): the kernels that took over from smart_fast_plain"; echo
  echo "1e5 LHS samples x hourly 10 yr + 1 yr warm-up, objective functions fused, no discharge matrix (tools/debug/reports_only.py);"
  echo "round 3 ran these through smart_fast_plain, the general step loop, unsliced: 24.8 ms (raw, gap 24) and 43.7 ms (gap 1)."
  for t in raw_gap24 raw_gap24_flat gap1; do echo; echo "## $t"; echo; sed -n '/kernel stats/,$p' profiles/r05_$t.md; done; } > profiles/r05_plain.md
cp gpurun_out/r05_recip_bits.txt profiles/
for f in r05_time_slice_soak r05_microbench_lanes r05_hook_time r05_config2_classes; do [ -f gpurun_out/$f.txt ] && cp gpurun_out/$f.txt profiles/; done
python tools/kernel_hashes.py > profiles/r05_kernel_hashes.txt 2>&1
for c in "" _c2 _c4 _c5 _c4shard; do grep '^{' gpurun_out/bench_r05$c.log > profiles/r05_bench_${c#_}.jsonl; done
mv profiles/r05_bench_.jsonl profiles/r05_bench_config3.jsonl   # (tools/gpu_final_check.sh's line is appended to it afterwards)
for k in steps intervals steps_every intervals_raw; do python tools/isa_report.py smart_fast_$k profiles/r05_isa_$k --hot > /dev/null; done
# (the listings of the two kernels with four instances of their loop each run to 1.5 MB of text: their block tables stay,
# the text is one command away -- python tools/isa_report.py smart_fast_steps_every /tmp/every --hot)
rm -f profiles/r05_isa_steps_every.s profiles/r05_isa_intervals_raw.s
python tools/kernel_resources.py > profiles/r05_kernel_resources.txt 2>/dev/null
for t in config3 flat_forcing flat_forcing_1e6 runs_of_6 raw_gap24 raw_gap24_flat gap1 config2 config4_1gpu config4_shard config5_1gpu; do
  echo "== $t"; grep -A4 "clock held and issue" profiles/r05_$t.md | tail -2; grep "timed steps\|traffic (corrected)" profiles/r05_$t.md; done
tail -1 profiles/r05_recip_bits.txt
cat gpurun_out/prof_r05_config3/source_hash.txt; python -c "import bench; print(bench.kernel_source_hash())"
