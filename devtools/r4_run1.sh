#!/bin/bash
mkdir -p gpurun_out/r4
L=gpurun_out/r4/run1.log; : > $L
timeout 900 python -m pytest tests/test_gemm_split16_gpu.py -x -q -m gpu 2>&1 | tail -15 >> $L
timeout 300 python devtools/bench_split16.py 50 2>&1 | grep -v amdgpu.ids >> $L
cat $L
