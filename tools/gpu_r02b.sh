#!/bin/bash
# round-2 session b: new bench.py (N=1), rocprofv3 kernel stats of it, PMC counters of the group workload
TAG=${1:-r02b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest (feeder + groups only)"
timeout 600 python -m pytest tests/test_batch_gpu.py -m gpu -q -x --timeout=300 -k "feeder or group" 2>&1 | tail -3
echo "== bench"
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 6000 $OUT/bench.json; tail -3 $OUT/bench.err
echo "== driver-shaped bench"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-variants --no-cpu-baseline > $OUT/bench_driver.json 2> $OUT/bench_driver.err
python3 -c "import json;j=json.load(open('$OUT/bench_driver.json'));print(j['value'],j['ms_per_step'],j['roofline']['frac'],j['parity_spot'])"
echo "== rocprofv3 kernel trace of the bench command"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-variants --no-spot > $GRAFT_REPO_ROOT/$OUT/prof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/prof.err
cd $GRAFT_REPO_ROOT
for f in $(find $OUT/prof -name "*kernel_stats*.csv" | head -1); do head -12 $f; cp $f $OUT/kernel_stats.csv; done
echo "== PMC"
bash tools/pmc_group.sh $TAG/pmc 1024 8 optimized > $OUT/pmc.log 2>&1; tail -120 $OUT/pmc.log
