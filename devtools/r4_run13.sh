#!/bin/bash
mkdir -p gpurun_out/r4
L=gpurun_out/r4/run13.log; : > $L
timeout 900 python -m pytest tests/test_ab_switches_gpu.py tests/test_tools_gpu.py tests/test_warpctc_gpu.py -x -q -m gpu 2>&1 | tail -5 >> $L
run() { echo "=== $*" >> $L; env "$@" timeout 600 python bench.py --steps 300 --warmup 50 --headline-only --no-gemm-profile $EXTRA 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('value', d['value'], 'ms', d['ms_per_step'])
" >> $L; }
EXTRA="" run A=1
EXTRA="--no-update-overlap" run A=1
EXTRA="" run ASLP_LOWEST_UPDATE_ON_SIDE=1
EXTRA="" run A=1
cat $L
