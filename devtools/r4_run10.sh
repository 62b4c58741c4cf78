#!/bin/bash
mkdir -p gpurun_out/r4
L=gpurun_out/r4/run10.log; : > $L
timeout 900 python -m pytest tests/test_gemm_split16_gpu.py tests/test_nnet_gpu.py tests/test_fullsize_gpu.py tests/test_kernels_gpu.py -x -q -m gpu 2>&1 | tail -12 >> $L
for v in 1 0; do echo "== SPLIT=$v" >> $L; ASLP_GEMM_SPLIT_F16=$v GEMM_PROFILE=1 timeout 300 python devtools/bench_cfg1.py 256 300 2>&1 | grep -v "amdgpu.ids\|^LOG" >> $L; done
cat $L
