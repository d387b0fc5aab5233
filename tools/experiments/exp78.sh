#!/bin/bash
# mix kernel: R rows loaded with the non-temporal hint (streamed once; keep X and the sibling pass's lines in L2?)
OUT=$GRAFT_REPO_ROOT/gpurun_out/s78; mkdir -p $OUT
export TMPDIR=/tmp
for N in 1024 4096 2048; do
python tools/sweep.py --clients $N --rates 5 --modes optimized --steps 200 2>&1 | grep optimized
done
bash tools/pmc_traffic.sh s78 > $OUT/pmc.log 2>&1; python3 -c "
import json; d=json.load(open('$OUT/pmc_latest.json')); print(d['hbm_bytes_per_block_polyphase'], {k:v['hbm_bytes'] for k,v in d['polyphase_kernels'].items()})"
