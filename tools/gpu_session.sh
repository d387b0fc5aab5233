#!/bin/bash
# tools/gpu_session.sh -- one gpurun call: GPU tests, microbench, bench, rocprofv3 kernel stats.
# Usage (from the build container): gpurun --timeout 1500 -- 'bash tools/gpu_session.sh [tag]'
TAG=${1:-s1}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== rocminfo" > $OUT/env.log
(rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -8; nproc; grep -m1 "model name" /proc/cpuinfo) >> $OUT/env.log 2>&1
echo "== pytest -m gpu"
timeout 900 python -m pytest tests -m gpu -q -x --timeout=600 > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -15 $OUT/pytest_gpu.log
echo "== ubench"
timeout 120 ./sdr-server_amd/build/ubench_valu > $OUT/ubench.log 2>&1
cat $OUT/ubench.log
echo "== bench"
timeout 600 python bench.py --steps 100 --warmup 10 > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json; tail -5 $OUT/bench.err
echo "== sweep"
timeout 600 python tools/sweep.py > $OUT/sweep.log 2>&1
cat $OUT/sweep.log
echo "== bench native"
timeout 300 python bench.py --steps 50 --warmup 5 --mode native --no-cpu-baseline > $OUT/bench_native.json 2> $OUT/bench_native.err
cat $OUT/bench_native.json; tail -3 $OUT/bench_native.err
echo "== rocprofv3 kernel trace"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-variants > $GRAFT_REPO_ROOT/$OUT/prof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/prof.err
cd $GRAFT_REPO_ROOT
tail -3 $OUT/prof.err
find $OUT/prof -type f | head; for f in $(find $OUT/prof -name "*kernel_stats*.csv" | head -1); do head -12 $f; done
