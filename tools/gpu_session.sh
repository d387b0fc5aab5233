#!/bin/bash
# tools/gpu_session.sh <tag> -- one gpurun call: GPU tests, smoke, bench (defaults + driver-shaped), rocprofv3 kernel stats of
# the bench command, PMC counters of the bench workload, kernel timeline, sweeps, per-config numbers.
# Usage (from the build container): gpurun --timeout 2400 -- 'bash tools/gpu_session.sh r02m'
TAG=${1:-s1}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
(rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -8; nproc; grep -m1 "model name" /proc/cpuinfo) > $OUT/env.txt 2>&1
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q -x --timeout=600 > $OUT/pytest_gpu.txt 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.txt
tail -5 $OUT/pytest_gpu.txt
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee $OUT/smoke.txt
echo "== bench"
timeout 900 python bench.py > $OUT/bench_n1.json 2> $OUT/bench.err
python3 -c "
import json;j=json.load(open('$OUT/bench_n1.json'))
print(j['value'],j['ms_per_step'],j['config']['us_per_block'],j['roofline']['frac'],j['roofline'].get('hbm_counter_frac'),j['parity_spot']['ok'], j['cpu_baseline']['value'])
print('native',j['native']['value'], j['native']['us_per_block'], j['native']['parity_spot']['ok'])
for k,v in j['variants'].items(): print(' ',k,v['value'],v['us_per_block'])"
echo "== driver-shaped bench"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_shape.json 2> $OUT/bench_driver.err
python3 -c "import json;j=json.load(open('$OUT/bench_driver_shape.json'));print(j['value'],j['ms_per_step'],j['steps'],j['warmup'])"
echo "== rocprofv3 kernel stats of the bench command"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-variants --no-spot > $GRAFT_REPO_ROOT/$OUT/prof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/prof.err
cd $GRAFT_REPO_ROOT
for f in $(find $OUT/prof -name "*kernel_stats*.csv" | head -1); do head -8 $f; cp $f $OUT/rocprofv3_kernel_stats.csv; done
for f in $(find $OUT/prof -name "*kernel_trace.csv" | head -1); do python3 tools/timeline.py $f 20 3 > $OUT/timeline_bench_1024clients.txt; tail -3 $OUT/timeline_bench_1024clients.txt; done
rm -rf $OUT/prof
echo "== PMC"
bash tools/pmc_group.sh $TAG/pmc 1024 8 optimized > $OUT/pmc.log 2>&1; tail -4 $OUT/pmc.log
bash tools/pmc_group.sh $TAG/pmc_native 1024 8 native > $OUT/pmc_native.log 2>&1; tail -4 $OUT/pmc_native.log
find $OUT/pmc $OUT/pmc_native -name "*.csv" -delete
echo "== chain stats"
XL_EXP_CHAIN_STATS=1 timeout 300 python tools/group_sweep.py --clients 128,256,512,1024,2048 --groups 8 --modes optimized --blocks 640 2>&1 | grep -v amdgpu.ids | tee $OUT/sweep_chain_stats.txt
echo "== sweeps"
timeout 600 python tools/group_sweep.py --clients 128,1024,4096 --groups 1,2,4,8 --modes optimized --poly3 --blocks 320 2>&1 | grep -v amdgpu.ids | tee $OUT/sweep_groups.txt
timeout 300 python tools/group_sweep.py --clients 1024 --groups 8 --modes optimized --m 128,256 --poly3 --blocks 320 2>&1 | grep -v amdgpu.ids | tee $OUT/sweep_m.txt
echo "== configs"
timeout 900 python tools/measure_configs.py 2>/dev/null > $OUT/configs.json; head -c 1500 $OUT/configs.json
echo "== ubench"
timeout 60 ./sdr-server_amd/build/ubench_chain7 2>&1 | tee $OUT/ubench_chain7.txt
timeout 60 ./sdr-server_amd/build/ubench_cumask 2>&1 | head -8 | tee $OUT/ubench_cumask.txt
