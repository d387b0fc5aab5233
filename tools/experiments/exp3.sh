#!/bin/bash
OUT=gpurun_out/s3; mkdir -p $OUT
echo "== baseline"; python tools/sweep.py --clients 64,1024,4096 --rates 5,1 2>&1 | grep -v amdgpu.ids | tee $OUT/base.log
echo "== YFAST"; XL_EXP_YFAST=1 python tools/sweep.py --clients 64,1024,4096 --rates 5,1 2>&1 | grep -v amdgpu.ids | tee $OUT/yfast.log
echo "== SAME_TAPS"; XL_EXP_SAME_TAPS=1 python tools/sweep.py --clients 64,1024,4096 --rates 5,1 2>&1 | grep -v amdgpu.ids | tee $OUT/same.log
echo "== SAME_TAPS+YFAST"; XL_EXP_SAME_TAPS=1 XL_EXP_YFAST=1 python tools/sweep.py --clients 1024 --rates 5 2>&1 | grep -v amdgpu.ids | tee $OUT/same_yfast.log
