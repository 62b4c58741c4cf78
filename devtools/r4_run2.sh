#!/bin/bash
mkdir -p gpurun_out/r4
L=gpurun_out/r4/run2.log; : > $L
timeout 1200 python -m pytest tests/test_gemm_split16_gpu.py tests/test_nnet_gpu.py tests/test_fullsize_gpu.py tests/test_kernels_gpu.py -x -q -m gpu 2>&1 | tail -15 >> $L
for v in 1 0; do
echo "=== ASLP_GEMM_SPLIT_F16=$v" >> $L
ASLP_GEMM_SPLIT_F16=$v timeout 600 python bench.py --steps 200 --warmup 50 --no-cfg3 --no-e2e-tool --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('value', d['value'], 'ms', d['ms_per_step'], 'xent', d['config']['avg_xent_per_frame']); print(json.dumps(d.get('gemm_all',{}).get('variants')))
    else: print(l)
" >> $L
done
cat $L
