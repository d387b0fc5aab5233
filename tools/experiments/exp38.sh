#!/bin/bash
CLIENTS=4096 RATES=5 bash tools/pmc.sh s38 XL_EXP_POLY=1 2>&1 | grep -v "amdgpu.ids" | grep "xlp_mix\|counters" 
