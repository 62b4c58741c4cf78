#!/bin/bash
mkdir -p gpurun_out/r4
L=gpurun_out/r4/run6.log; : > $L
for i in 1 2 3; do
timeout 600 python bench.py --steps 300 --warmup 50 --no-cfg3 --no-e2e-tool --no-cpu-baseline --no-gemm-profile 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('value', d['value'], 'ms', d['ms_per_step'], 'xent', d['config']['avg_xent_per_frame'])
" >> $L
done
bash devtools/prof_cfg2_timeline.sh >> $L 2>&1
cp gpurun_out/cfg2_timeline.txt gpurun_out/r4/cfg2_timeline_stage4.txt
cat gpurun_out/r4/cfg2_timeline_stage4.txt >> $L
cat $L
