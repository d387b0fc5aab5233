#!/bin/bash
# round-2 session c: side-stream NCO chain (parity + timing), C multi host selftest, reference callers, new bench
TAG=${1:-r02c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q -x --timeout=600 > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -40 $OUT/pytest_gpu.log
echo "== group sweep: side stream vs NCO role"
for side in 0 1; do
  XL_EXP_NCO_SIDE=$side timeout 600 python tools/group_sweep.py --clients 128,1024,4096 --groups 1,8 --modes optimized --poly3 2>&1 | grep -v amdgpu.ids | sed "s/^/side=$side /" | tee -a $OUT/group_sweep_side.txt
done
XL_EXP_NCO_SIDE=1 timeout 300 python tools/group_sweep.py --clients 128,1024 --groups 8 --modes native 2>&1 | grep -v amdgpu.ids | sed "s/^/side=1 /" | tee -a $OUT/group_sweep_side.txt
XL_EXP_NCO_SIDE=0 timeout 300 python tools/group_sweep.py --clients 128,1024 --groups 8 --modes native 2>&1 | grep -v amdgpu.ids | sed "s/^/side=0 /" | tee -a $OUT/group_sweep_side.txt
echo "== bench (C host)"
timeout 900 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python3 -c "
import json;j=json.load(open('$OUT/bench.json'))
print(j['value'],j['ms_per_step'],j['config']['us_per_block'],j['config']['feed'],j['roofline']['frac'],j['roofline']['kernels_ms'],j['parity_spot'])
print('native',j['native'])
for k,v in j['variants'].items(): print(k,v['value'],v['us_per_block'],v['plan'])
"; tail -3 $OUT/bench.err
echo "== bench (torch feeder)"
timeout 600 python bench.py --feed torch --no-cpu-baseline --no-variants > $OUT/bench_torch.json 2> $OUT/bench_torch.err
python3 -c "
import json;j=json.load(open('$OUT/bench_torch.json'));print(j['value'],j['ms_per_step'],j['config']['feed'],j['parity_spot'])"; tail -3 $OUT/bench_torch.err
echo "== feed over RCCL selftest"
timeout 300 python tools/feed_nccl_selftest.py 2>&1 | grep -v amdgpu.ids | tail -3
