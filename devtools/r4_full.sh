#!/bin/bash
mkdir -p gpurun_out/r4
L=gpurun_out/r4/full.log; : > $L
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 >> $L
timeout 600 python bench.py --steps 200 --warmup 50 --no-cfg3 --no-e2e-tool --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('value', d['value'], 'ms', d['ms_per_step'], 'xent', d['config']['avg_xent_per_frame']); print(json.dumps(d.get('gemm_all',{}).get('variants')))
" >> $L
cat $L
