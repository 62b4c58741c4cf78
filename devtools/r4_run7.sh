#!/bin/bash
mkdir -p gpurun_out/r4
L=gpurun_out/r4/run7.log; : > $L
timeout 900 python -m pytest tests/test_gemm_split16_gpu.py tests/test_fullsize_gpu.py tests/test_nnet_gpu.py -x -q -m gpu 2>&1 | tail -12 >> $L
for k in 1 0; do
echo "=== ASLP_GEMM_KS128=$k" >> $L
ASLP_GEMM_KS128=$k timeout 300 python devtools/bench_split16.py 100 2>&1 | grep -A1 "^   product\|TN" | grep -B1 "^TN" >> $L
ASLP_GEMM_KS128=$k timeout 600 python bench.py --steps 300 --warmup 50 --no-cfg3 --no-e2e-tool --no-cpu-baseline --no-gemm-profile 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('value', d['value'], 'ms', d['ms_per_step'], 'xent', d['config']['avg_xent_per_frame'])
" >> $L
done
cat $L
