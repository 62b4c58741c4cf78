#!/bin/bash
mkdir -p gpurun_out/r4
L=gpurun_out/r4/run8.log; : > $L
timeout 900 python -m pytest tests/test_gemm_split16_gpu.py -x -q -m gpu 2>&1 | tail -4 >> $L
for pad in 64 0 128 32; do
echo "=== ASLP_S16_LD_PAD=$pad" >> $L
ASLP_S16_LD_PAD=$pad timeout 300 python devtools/bench_split16.py 100 2>&1 | grep "product alone" | head -3 | tr '\n' ' ' >> $L
echo >> $L
ASLP_S16_LD_PAD=$pad timeout 600 python bench.py --steps 300 --warmup 50 --no-cfg3 --no-e2e-tool --no-cpu-baseline --no-gemm-profile 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('value', d['value'], 'ms', d['ms_per_step'], 'xent', d['config']['avg_xent_per_frame'])
" >> $L
done
cat $L
