#!/bin/bash
# drop-in filter with ping-pong sample images (no history-move launch): parity (incl. the randomised sequences, 60 seeds) + latency
XL_TEST_FUZZ_SEEDS=60 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -2
echo "zero-copy:"; python tools/measure_dropin.py 2>/dev/null | tail -1
