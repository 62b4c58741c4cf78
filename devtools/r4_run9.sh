#!/bin/bash
mkdir -p gpurun_out/r4
L=gpurun_out/r4/run9.log; : > $L
timeout 1500 python -m pytest tests/test_bench_gpu.py tests/test_gemm_split16_gpu.py -x -q -m gpu 2>&1 | tail -30 >> $L
timeout 900 python bench.py > gpurun_out/r4/bench_full.json 2> gpurun_out/r4/bench_full.err
echo "bench rc $?" >> $L
python - >> $L <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r4/bench_full.json') if l.startswith('{')][-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'cold', d.get('cold_value'))
print('roofline', json.dumps(d['roofline']))
print('fp32_instruction', json.dumps(d.get('fp32_instruction')))
pa=d.get('product_accuracy',{})
print('accuracy all_le', pa.get('split_le_fp32'))
for r in pa.get('shapes',[]): print('  ', r)
for k,v in d.get('hbm_kernels',{}).get('kernels',{}).items(): print('  hbm', k, round(v['us'],1), 'us', round(v['gb_per_s']), 'GB/s', round(v['frac_of_hbm_peak'],3))
print('cfg1_gpu', json.dumps(d.get('cfg1_gpu')))
c3=d.get('cfg3',{})
print('cfg3 chunked', json.dumps(c3.get('chunked_xent')))
print('cfg3 ctc', json.dumps(c3.get('whole_utterance_warpctc')))
print('e2e', json.dumps(d.get('e2e_tool')))
print('cpu', json.dumps(d.get('cpu_baseline')))
PY
tail -5 gpurun_out/r4/bench_full.err >> $L
cat $L
