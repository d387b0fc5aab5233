#!/bin/bash
# tools/gpu_session.sh -- one gpurun call: GPU tests, smoke, bench (+ native), rocprofv3 kernel stats, PMC traffic, sweeps.
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
tail -5 $OUT/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee $OUT/smoke.log
echo "== bench"
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json; tail -3 $OUT/bench.err
echo "== bench native"
timeout 300 python bench.py --steps 50 --warmup 5 --mode native --no-cpu-baseline --no-variants > $OUT/bench_native.json 2> $OUT/bench_native.err
cat $OUT/bench_native.json; tail -3 $OUT/bench_native.err
echo "== rocprofv3 kernel trace of the bench command"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-variants > $GRAFT_REPO_ROOT/$OUT/prof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/prof.err
cd $GRAFT_REPO_ROOT
tail -2 $OUT/prof.err
for f in $(find $OUT/prof -name "*kernel_stats*.csv" | head -1); do head -12 $f; done
echo "== PMC traffic"
bash tools/pmc_traffic.sh $TAG > $OUT/pmc_traffic.log 2>&1; tail -5 $OUT/pmc_traffic.log
echo "== sweeps"
timeout 600 python tools/sweep.py --clients 256,512,1024,2048,4096 --rates 5,1 --modes optimized,native --steps 100 2>&1 | grep -v amdgpu.ids > $OUT/sweep.log
cat $OUT/sweep.log
timeout 600 python tools/poly_check.py 128,256,512,1024,2048,4096 5 2>&1 | grep -v amdgpu.ids | grep -v "^   " > $OUT/poly_check.log
cat $OUT/poly_check.log
timeout 300 python tools/feed_overhead.py 2>&1 | grep -v amdgpu.ids | tee $OUT/feed_overhead.log
