#!/bin/bash
# tools/gpu_r06.sh <tag> <steps...> -- ONE parametrised GPU session script (replaces the per-session gpu_r05*.sh files).
# Usage (build container): gpurun --timeout 1800 -- 'bash tools/gpu_r06.sh r06a tests smoke bench'
# Steps: tests | smoke | bench | stats | stats5 | stats4096 | configs | custom:<script under tools/>   Outputs: gpurun_out/<tag>/
TAG=${1:-r06}; shift; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
(rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -8; nproc; grep -m1 "model name" /proc/cpuinfo; /opt/rocm/bin/hipcc --version | head -2) > $OUT/env.txt 2>&1
stats() {  # stats <name> <command...>: rocprofv3 --kernel-trace --stats of a command; keeps the kernel_stats csv
  local name=$1; shift
  (cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -o p -- "$@" > $OUT/prof_$name.log 2>&1)
  for f in $(find $OUT/prof_$name -name "*kernel_stats*.csv" | head -1); do head -8 $f; cp $f $OUT/rocprofv3_kernel_stats_$name.csv; done
  for f in $(find $OUT/prof_$name -name "*kernel_trace.csv" | head -1); do python3 tools/timeline.py $f 20 3 > $OUT/timeline_$name.txt 2>/dev/null; done
  rm -rf $OUT/prof_$name
}
for step in "$@"; do
  echo "== $step"
  case $step in
    tests)  ( time timeout 1200 python -m pytest tests -m gpu -q --timeout=600 --durations=10 > $OUT/pytest_gpu.txt 2>&1 ) 2>&1 | grep real
            tail -4 $OUT/pytest_gpu.txt ;;
    testsx) timeout 1200 python -m pytest tests -m gpu -q -x --timeout=600 > $OUT/pytest_gpu.txt 2>&1; tail -15 $OUT/pytest_gpu.txt ;;
    smoke)  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee $OUT/smoke.txt | tail -3 ;;
    bench)  ( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_stdout.txt 2> $OUT/bench.err ) 2>&1 | grep real
            tail -3 $OUT/bench.err; wc -c $OUT/bench_stdout.txt; tail -1 $OUT/bench_stdout.txt
            cp profiles/bench_last_full.json $OUT/bench_full.json 2>/dev/null ;;
    stats)  stats bench python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-variants --no-spot --no-pmc ;;
    stats5) stats config5_1024clients python $GRAFT_REPO_ROOT/tools/group_sweep.py --shape config5 --clients 1024 --groups 8 --modes optimized --blocks 640
            grep optimized $OUT/prof_config5_1024clients.log ;;
    stats4096) stats 4096clients python $GRAFT_REPO_ROOT/tools/group_sweep.py --clients 4096 --groups 8 --modes optimized --blocks 320
            grep optimized $OUT/prof_4096clients.log ;;
    configs) timeout 900 python tools/measure_configs.py 2>/dev/null > $OUT/configs.json; head -c 3000 $OUT/configs.json ;;
    custom:*) bash tools/${step#custom:} $OUT 2>&1 | tail -60 ;;
    *) echo "unknown step $step" ;;
  esac
done
