#!/bin/bash
OUT=$1
timeout 600 python bench.py --no-cpu-baseline --no-pmc --full-json $OUT/bench_nopmc_full.json 2>/dev/null | tail -1 | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print(j['value'], {k:(v['value'],v['us_per_block']) for k,v in j['configs'].items()})"
