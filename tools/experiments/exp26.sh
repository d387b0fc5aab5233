#!/bin/bash
# NCO-role workgroups with 1/2/4 client-carrying waves (balance the CUs that host them)
OUT=gpurun_out/s26; mkdir -p $OUT
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q -x --timeout=600 2>&1 | tail -3 | tee $OUT/pytest.log
for w in 1 2 4; do echo "== wpw $w"; XL_EXP_NCOWPW=$w python tools/sweep.py --clients 512,960,1024,2048 --rates 5 --modes optimized,native --steps 100 2>&1 | grep -v amdgpu.ids | tee -a $OUT/sweep_wpw$w.log; done
for w in 4 2; do echo "== trace wpw $w"; XL_EXP_NCOWPW=$w XL_EXP_TRACE=$OUT/t.bin python tools/sweep.py --clients 1024 --rates 5 --modes optimized --steps 3 2>&1 | grep -v amdgpu.ids | tail -1; python tools/trace_analyze.py $OUT/t.bin | tee $OUT/trace_wpw$w.txt; rm -f $OUT/t.bin; done
