#!/bin/bash
mkdir -p gpurun_out/r4
L=gpurun_out/r4/run15.log; : > $L
timeout 1500 python -m pytest tests/test_gemm_split16_gpu.py tests/test_nnet_gpu.py tests/test_fullsize_gpu.py tests/test_gradcheck_gpu.py tests/test_components_gpu.py -x -q -m gpu 2>&1 | tail -12 >> $L
for v in 1 0; do echo "== SPLIT=$v" >> $L; ASLP_GEMM_SPLIT_F16=$v timeout 300 python devtools/bench_cfg1.py 256 400 2>&1 | grep "cfg1 minibatch" >> $L; done
bash devtools/prof_cfg1_timeline.sh > /dev/null 2>&1; cut -c1-130 gpurun_out/cfg1_timeline.txt >> $L
timeout 600 python bench.py --steps 300 --warmup 50 --headline-only --no-gemm-profile 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('cfg2 value', d['value'], 'ms', d['ms_per_step'])
" >> $L
cat $L
