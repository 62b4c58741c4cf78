#!/bin/bash
mkdir -p gpurun_out/r4
L=gpurun_out/r4/run14.log; : > $L
run() { echo "=== $*" >> $L; env "$@" timeout 600 python bench.py --steps 300 --warmup 50 --headline-only --no-gemm-profile 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('value', d['value'], 'ms', d['ms_per_step'])
" >> $L; }
run ASLP_GEMM_KS128_COST=1.6
run ASLP_GEMM_KS128_COST=1.4
run ASLP_GEMM_KS128_COST=1.6
run ASLP_GEMM_KS128_COST=1.4
for c in 1.6 1.4; do ASLP_GEMM_KS128_COST=$c timeout 300 python devtools/bench_split16.py 100 2>&1 | grep -B1 "TN  3000" | head -2 >> $L; done
cat $L
