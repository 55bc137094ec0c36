#!/bin/bash
# condense what tools/gpu_profile_r03.sh brought back into profiles/ (run in the build container, after the gpurun call)
S=tools/summarize_profile.py
# the instruction models first: the summaries quote their fp64 shares (needs profiles/r03_fill_paths.json: tools/fill_paths.py)
python tools/isa_model.py intervals profiles/r03_isa_model_intervals > /dev/null 2>&1
python tools/isa_model.py steps profiles/r03_isa_model_steps > /dev/null 2>&1
python $S gpurun_out/prof_r03_config3 profiles/r03_config3 "config3:runs_per_gpu=100000:discharge=1:math=fast" > /dev/null
python $S gpurun_out/prof_r03_config4_shard profiles/r03_config4_shard "config4:runs_per_gpu=125000:discharge=0:math=fast" > /dev/null
python $S gpurun_out/prof_r03_config4_1gpu profiles/r03_config4_1gpu "config4:runs_per_gpu=1000000:discharge=0:math=fast" > /dev/null
python $S gpurun_out/prof_r03_config5_1gpu profiles/r03_config5_1gpu "config5:runs_per_gpu=640000:discharge=0:math=fast" > /dev/null
python $S gpurun_out/prof_r03_config2 profiles/r03_config2 "config2:runs_per_gpu=10000:discharge=1:math=fast" > /dev/null
python $S gpurun_out/prof_r03_flat_forcing profiles/r03_flat_forcing > /dev/null
python $S gpurun_out/prof_r03_flat_forcing_1e6 profiles/r03_flat_forcing_1e6 > /dev/null
python $S gpurun_out/prof_r03_runs_of_6 profiles/r03_runs_of_6 > /dev/null
cp gpurun_out/r03_steps_bits.txt gpurun_out/r03_recip_bits.txt gpurun_out/r03_time_slice_soak.txt profiles/
for c in "" _c2 _c4 _c5 _c4shard; do grep '^{' gpurun_out/bench_r03$c.log > profiles/r03_bench_${c#_}.jsonl; done
mv profiles/r03_bench_.jsonl profiles/r03_bench_config3.jsonl
for k in steps intervals intervals_exits; do python tools/isa_report.py smart_fast_$k profiles/r03_isa_$k --hot > /dev/null; done
python tools/isa_report.py smart_fast_illcond /tmp/r03_isa_illcond --hot > /dev/null && cp /tmp/r03_isa_illcond.json profiles/r03_isa_illcond.json
python tools/kernel_resources.py > profiles/r03_kernel_resources.txt 2>/dev/null
for t in config3 flat_forcing flat_forcing_1e6 runs_of_6 config2 config4_1gpu config4_shard config5_1gpu; do
  echo "== $t"; grep -A4 "clock held and issue" profiles/r03_$t.md | tail -2; grep "timed steps\|traffic (corrected)" profiles/r03_$t.md; done
tail -1 profiles/r03_recip_bits.txt; grep "differ" profiles/r03_steps_bits.txt; grep -c "0 differ, 0 with" profiles/r03_time_slice_soak.txt
cat gpurun_out/prof_r03_config3/source_hash.txt; python -c "import bench; print(bench.kernel_source_hash())"
