#!/bin/bash
# drop-in filter: kernels read the pinned input / write the pinned output over PCIe (default) vs staged copies
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -2
echo "zero-copy:"; python tools/measure_dropin.py 2>/dev/null | tail -1
echo "staged copies:"; XL_EXP_DROPIN_COPY=1 python tools/measure_dropin.py 2>/dev/null | tail -1
