#!/bin/bash
mkdir -p gpurun_out/r3
L=gpurun_out/r3/run7.log; : > $L
timeout 1200 python -m pytest tests/test_nnet_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu 2>&1 | tail -8 >> $L
echo "=== bench default-like (driver command)" >> $L
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3/run7_bench.json 2>gpurun_out/r3/run7_bench.err
echo "rc=$?" >> $L
python - <<'PY' >> $L
import json
d=json.load(open('gpurun_out/r3/run7_bench.json'))
print({k: d[k] for k in ('value','cold_value','ms_per_step','prewarm_steps')})
print('roofline', d['roofline']['frac'], 'gemm_all', d['gemm_all']['tflops'])
print('cfg3', d['cfg3']['chunked_xent']['ms_per_step'], d['cfg3']['chunked_xent']['frac_of_mfma_peak'], d['cfg3']['whole_utterance_warpctc']['ms_per_step'], d['cfg3']['whole_utterance_warpctc']['frac_of_mfma_peak'])
print('cfg1_gpu', d.get('cfg1_gpu'))
print('e2e_tool', d.get('e2e_tool'))
print('recurrent_layers', d.get('recurrent_layers'))
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['kind'])
PY
cat $L
