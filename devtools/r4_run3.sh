#!/bin/bash
mkdir -p gpurun_out/r4
L=gpurun_out/r4/run3.log; : > $L
timeout 600 python -m pytest tests/test_gemm_split16_gpu.py tests/test_fullsize_gpu.py tests/test_nnet_gpu.py -x -q -m gpu 2>&1 | tail -5 >> $L
bash devtools/prof_cfg2_timeline.sh >> $L 2>&1
cp gpurun_out/cfg2_timeline.txt gpurun_out/r4/cfg2_timeline_stage1.txt
cat gpurun_out/r4/cfg2_timeline_stage1.txt >> $L
cat $L
