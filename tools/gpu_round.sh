#!/bin/bash
# The GPU sessions of a round, one script (round 3 had gpu_r03_a.sh ... gpu_r03_v.sh).  Run through gpurun:
#   gpurun --timeout N -- 'bash tools/gpu_round.sh <stage> [args]'
# stages
#   quick [-k EXPR]    the GPU tests selected by EXPR (default: the round's new ones), then tools/debug/gap1_hourly.py
#   suite              the whole GPU suite
#   bench [args]       bench.py with the driver's K / W (--steps 20 --warmup 5) and a one-line digest of the legs
#   ab LIB...          interleaved A/B of library variants (tools/ab_variants.sh)
#   bits LIB           tools/debug/steps_bits.py: every output of the merged kernels, variant LIB against the default build
#   profile TAG [bench args]   tools/profile.sh (kernel trace + PMC passes of the bench command)
#   phases SCRIPT ...  code-placement scan: the leg SCRIPT (e.g. tools/debug/flat_only.py 100000 6) under every phase variant
#   soak               tools/gpu_soak.sh: thousands of sliced launches against the unsliced one, competitors beside them
#   slices             4 / 8 / 16 time slices on configs 4 and 5 (where the hand-over is the only HBM traffic)
#   final              what the driver runs at round end (tools/gpu_final_check.sh)
# round 4, the pair blocks / the every-step stream / the wet interval's turns (logs: profiles/r04_ab_pair_blocks.txt, r04_ab_wet_turns.txt)
#   pairs              the pair blocks against the threaded chunks of the same library (SMART_PAIR_BLOCKS=0): bits, then the flat / raw legs both ways
#   every              the every-step stream against the step-by-step loop: its GPU tests, bits, the gap-1 and raw legs both ways
#   strides            block strides of the pair blocks (variants s*: build_variants.py sN=-DSMART_P_STRIDE=N) on the flat leg
#   slices-flat        slice counts 8 ... 48 on the flat-forcing leg
#   slices-headline    slice counts on the headline run and the run engine
#   slices-legs        the default slice count against 16 on every bench leg
#   phases-every       placement of the every-step loop (variants e*: -DSMART_EVERY_PHASE=N), then slices-flat
#   turns              library variants in tools/variants/ against the default on every bench leg, five interleaved rounds (the wet interval's turns, the lean report)
# Everything a stage prints also lands in gpurun_out/<stage>_*.log.
export TMPDIR=/tmp
mkdir -p gpurun_out
STAGE=${1:-quick}; shift
case $STAGE in
quick)
    EXPR=${2:-"raw_and_every or bench_legs or without_a_plan or bare_command"}
    timeout 2400 python -m pytest tests -m gpu -x -q -k "$EXPR" > gpurun_out/quick_pytest.log 2>&1; tail -15 gpurun_out/quick_pytest.log
    python tools/debug/gap1_hourly.py 2>&1 | tee gpurun_out/quick_gap1.log | tail -8
    ;;
suite)
    timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/suite_pytest.log 2>&1; tail -15 gpurun_out/suite_pytest.log
    ;;
bench)
    python bench.py --steps 20 --warmup 5 "$@" > gpurun_out/bench_round.log 2>&1
    grep '^{' gpurun_out/bench_round.log | python tools/bench_digest.py
    ;;
ab)
    bash tools/ab_variants.sh "$@" 2>&1 | tee gpurun_out/ab_round.log | tail -40
    ;;
bits)   # bits LIB: outputs of the merged kernels from library variant LIB and from the default build, bit for bit
    SMART_AMD_LIB=$PWD/$1 python tools/debug/steps_bits.py dump /tmp/bits_a.npz > /dev/null
    python tools/debug/steps_bits.py dump /tmp/bits_b.npz > /dev/null
    python tools/debug/steps_bits.py compare /tmp/bits_a.npz /tmp/bits_b.npz 2>&1 | tee gpurun_out/bits_$(basename $1 .so).log | tail -5
    ;;
profile)
    bash tools/profile.sh "$@"
    ;;
phases) # phases SCRIPT [args]: every tools/variants/libsmart_amd_p<N>.so (build_variants.py pN=-DSMART_..._PHASE=N) on one leg
    for rep in 1 2; do for f in $(ls tools/variants/libsmart_amd_p*.so | sort -V); do
        echo -n "$(basename $f .so | sed s/libsmart_amd_//): "; SMART_AMD_LIB=$PWD/$f python "$@" 2>/dev/null | grep " ms" | sort -n | head -3 | tr '\n' ' '; echo
    done; done 2>&1 | tee gpurun_out/phases_round.log
    ;;
soak)
    bash tools/gpu_soak.sh 2>&1 | tee gpurun_out/soak_round.log | tail -40
    ;;
slices) # slice counts at the loads where the hand-over is the only HBM traffic (configs 4 and 5 on one GPU)
    for cfg in 4 5; do for rep in 1 2; do for k in default 4 8 16; do
        if [ $k = default ]; then unset SMART_TIME_SLICES; else export SMART_TIME_SLICES=$k; fi
        echo -n "config $cfg slices $k: "; timeout 600 python bench.py --config $cfg --steps 4 --warmup 1 --no-cpu-baseline --no-flat 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms  %s' % (d['roofline']['launch_ms'], d['roofline']['kernel']))"
    done; done; done 2>&1 | tee gpurun_out/slices_round.log
    ;;
final)
    bash tools/gpu_final_check.sh
    ;;
pairs)
    python tools/debug/steps_bits.py dump /tmp/bits_new.npz > /dev/null || echo "dump (pairs) failed"
    SMART_PAIR_BLOCKS=0 python tools/debug/steps_bits.py dump /tmp/bits_old.npz > /dev/null || echo "dump (threaded) failed"
    python tools/debug/steps_bits.py compare /tmp/bits_new.npz /tmp/bits_old.npz 2>&1 | tail -4
    for rep in 1 2; do
      echo -n "pairs    flat: "; python tools/debug/flat_only.py 100000 8 2>/dev/null | grep " ms" | sort -n | head -3 | tr '\n' ' '; echo
      echo -n "threaded flat: "; SMART_PAIR_BLOCKS=0 python tools/debug/flat_only.py 100000 8 2>/dev/null | grep " ms" | sort -n | head -3 | tr '\n' ' '; echo
    done
    echo -n "pairs    raw_flat: "; python tools/debug/reports_only.py raw_flat 2>/dev/null | grep " ms" | sort -n | head -3 | tr '\n' ' '; echo
    echo -n "threaded raw_flat: "; SMART_PAIR_BLOCKS=0 python tools/debug/reports_only.py raw_flat 2>/dev/null | grep " ms" | sort -n | head -3 | tr '\n' ' '; echo
    echo -n "pairs    flat 1e6: "; python tools/debug/flat_only.py 1000000 4 2>/dev/null | grep " ms" | sort -n | head -3 | tr '\n' ' '; echo
    echo -n "threaded flat 1e6: "; SMART_PAIR_BLOCKS=0 python tools/debug/flat_only.py 1000000 4 2>/dev/null | grep " ms" | sort -n | head -3 | tr '\n' ' '; echo
    ;;
every)
    timeout 1500 python -m pytest tests -m gpu -x -q -k "every or pair_blocks or raw_and or bench_legs" 2>&1 | tail -4
    python tools/debug/steps_bits.py dump /tmp/bits_new.npz > /dev/null || echo "dump failed"
    SMART_PAIR_BLOCKS=0 python tools/debug/steps_bits.py dump /tmp/bits_old.npz > /dev/null || echo "dump (old) failed"
    python tools/debug/steps_bits.py compare /tmp/bits_new.npz /tmp/bits_old.npz 2>&1 | tail -3
    for rep in 1 2; do
      echo -n "stream every: "; python tools/debug/reports_only.py every 100000 6 2>/dev/null | grep " ms" | sort -n | head -3 | tr '\n' ' '; echo
      echo -n "step-by-step every: "; SMART_PAIR_BLOCKS=0 python tools/debug/reports_only.py every 100000 6 2>/dev/null | grep " ms" | sort -n | head -3 | tr '\n' ' '; echo
      echo -n "pairs raw_flat: "; python tools/debug/reports_only.py raw_flat 100000 6 2>/dev/null | grep " ms" | sort -n | head -3 | tr '\n' ' '; echo
    done
    ;;
strides)
    for rep in 1 2 3; do
      for f in default $(ls tools/variants/libsmart_amd_s*.so); do
        if [ $f = default ]; then unset SMART_AMD_LIB; else export SMART_AMD_LIB=$PWD/$f; fi
        echo -n "$(basename $f .so | sed s/libsmart_amd_//): "; python tools/debug/flat_only.py 100000 8 2>/dev/null | grep " ms" | sort -n | head -3 | tr '\n' ' '; echo
      done
      unset SMART_AMD_LIB
      echo -n "threaded: "; SMART_PAIR_BLOCKS=0 python tools/debug/flat_only.py 100000 8 2>/dev/null | grep " ms" | sort -n | head -3 | tr '\n' ' '; echo
    done 2>&1 | tee gpurun_out/pairs_ab.log
    ;;
slices-flat)
    for rep in 1 2; do for k in 8 12 16 20 24 32 48; do
      echo -n "flat slices $k: "; SMART_TIME_SLICES=$k python tools/debug/flat_only.py 100000 6 2>/dev/null | grep " ms" | sort -n | head -3 | tr '\n' ' '; echo
    done; done 2>&1 | tee gpurun_out/slices_flat.log
    ;;
slices-headline)
    for rep in 1 2; do for k in 12 16 20 24 32; do
      echo -n "headline slices $k: "; SMART_TIME_SLICES=$k python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-flat --no-strong 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms' % d['roofline']['launch_ms'])"
      echo -n "runs6 slices $k: "; SMART_TIME_SLICES=$k python tools/debug/runs_only.py 100000 6 2>/dev/null | grep " ms" | sort -n | head -3 | tr '\n' ' '; echo
    done; done 2>&1 | tee gpurun_out/slices_headline.log
    ;;
slices-legs)
    for rep in 1 2 3; do for k in default 16; do
      if [ $k = default ]; then unset SMART_TIME_SLICES; else export SMART_TIME_SLICES=$k; fi
      echo -n "slices $k: "; python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-strong 2>/dev/null | tail -1 | python tools/bench_digest.py | grep " ms" | awk '{printf "%s %s | ", $1, $2}'; echo
    done; done 2>&1 | tee gpurun_out/slices_ab.log
    ;;
phases-every)
    for rep in 1 2; do for f in default $(ls tools/variants/libsmart_amd_e*.so); do
      if [ $f = default ]; then unset SMART_AMD_LIB; else export SMART_AMD_LIB=$PWD/$f; fi
      echo -n "$(basename $f .so | sed s/libsmart_amd_//) every: "; python tools/debug/reports_only.py every 100000 6 2>/dev/null | grep " ms" | sort -n | head -3 | tr '\n' ' '; echo
    done; done 2>&1 | tee gpurun_out/every_phase.log
    unset SMART_AMD_LIB
    bash tools/gpu_round.sh slices-flat
    ;;
turns)
    for rep in 1 2 3 4; do for f in default $(ls tools/variants/libsmart_amd_*.so); do
      if [ $f = default ]; then unset SMART_AMD_LIB; else export SMART_AMD_LIB=$PWD/$f; fi
      echo -n "$(basename $f .so | sed s/libsmart_amd_//): "; python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-strong 2>/dev/null | tail -1 | python tools/bench_digest.py | grep " ms" | awk '{printf "%s %s | ", $1, $2}'; echo
    done; done 2>&1 | tee gpurun_out/wet_ab.log
    ;;
icache) # instruction-cache requests / misses of the time-loop kernels (the pair blocks are 84 KB of code an instance)
    export TMPDIR=/tmp
    run() { tag=$1; shift
      rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVES --kernel-trace --output-format csv -d gpurun_out/icache_$tag -o pmc -- python3 "$@" > gpurun_out/icache_$tag.log 2>&1
      echo "== $tag ($*) rc=$?"; grep " ms" gpurun_out/icache_$tag.log | sort -n | head -2 | tr '\n' ' '; echo
      python3 tools/pmc_digest.py gpurun_out/icache_$tag; }
    ( run steps_1e6 tools/debug/flat_only.py 1000000 3
      run steps_1e5 tools/debug/flat_only.py 100000 6
      export SMART_PAIR_BLOCKS=0
      run threaded_1e6 tools/debug/flat_only.py 1000000 3
      run threaded_1e5 tools/debug/flat_only.py 100000 6
      unset SMART_PAIR_BLOCKS
      run intervals_1e6 bench.py --config 4 --steps 3 --warmup 1 --no-cpu-baseline --no-flat --no-strong
      run intervals_1e5 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-flat --no-strong ) 2>&1 | tee gpurun_out/icache.log
    ;;
*)
    echo "unknown stage $STAGE"; exit 2
    ;;
esac
