#!/bin/bash
mkdir -p gpurun_out/r3
L=gpurun_out/r3/run6.log; : > $L
timeout 900 python -m pytest tests/test_components_gpu.py tests/test_oracle_cpu.py -x -q 2>&1 | tail -30 >> $L
echo "=== rnn tests (default = chains of 8 + fast act)" >> $L
timeout 900 python -m pytest tests/test_rnn_gpu.py tests/test_cfg3_step_gpu.py tests/test_gradcheck_gpu.py tests/test_ab_switches_gpu.py tests/test_ref_blas_gpu.py tests/test_warpctc_gpu.py -x -q -m gpu 2>&1 | tail -8 >> $L
echo "=== bench lc default / exact act" >> $L
timeout 300 python devtools/bench_lc.py 32 100 2>&1 | tail -1 >> $L
ASLP_LSTM_FAST_ACT=0 timeout 300 python devtools/bench_lc.py 32 100 2>&1 | tail -1 >> $L
timeout 300 python devtools/bench_lc.py 32 100 2>&1 | tail -1 >> $L
cat $L
