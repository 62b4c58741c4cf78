#!/bin/bash
# round 3, GPU run 1: parity of the half-chain LSTM kernels, then A/B timing (half chains, both pairings, chains of 8)
mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests/test_rnn_gpu.py tests/test_cfg3_step_gpu.py tests/test_gradcheck_gpu.py tests/test_ab_switches_gpu.py tests/test_ref_blas_gpu.py -x -q -m gpu > gpurun_out/r3/run1_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3/run1_tests.log
for cfg in "1 1" "1 0" "0 1"; do
  set -- $cfg
  echo "=== HALF_CHAINS=$1 HALF_MAP=$2" >> gpurun_out/r3/run1_bench.log
  ASLP_LSTM_HALF_CHAINS=$1 ASLP_LSTM_HALF_MAP=$2 SEQ_TIMING=1 SEQ_CENSUS=$1 timeout 300 python devtools/bench_lc.py 32 100 >> gpurun_out/r3/run1_bench.log 2>&1
done
echo "=== again HALF=1 MAP=1 / HALF=0 (A-B-A)" >> gpurun_out/r3/run1_bench.log
ASLP_LSTM_HALF_CHAINS=1 timeout 300 python devtools/bench_lc.py 32 200 2>&1 | tail -1 >> gpurun_out/r3/run1_bench.log
ASLP_LSTM_HALF_CHAINS=0 timeout 300 python devtools/bench_lc.py 32 200 2>&1 | tail -1 >> gpurun_out/r3/run1_bench.log
ASLP_LSTM_HALF_CHAINS=1 timeout 300 python devtools/bench_lc.py 32 200 2>&1 | tail -1 >> gpurun_out/r3/run1_bench.log
tail -5 gpurun_out/r3/run1_tests.log; cat gpurun_out/r3/run1_bench.log | grep -v "^LOG"
