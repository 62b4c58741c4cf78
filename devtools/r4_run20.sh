#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
{
python3 -m pytest tests/test_dnn_ab_switches_gpu.py tests/test_gemm_split16_gpu.py tests/test_nnet_gpu.py tests/test_fullsize_gpu.py tests/test_kernels_gpu.py -x -q -m gpu 2>&1 | tail -6
for i in 1 2; do
python3 bench.py --steps 300 --warmup 50 --headline-only | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 bn planes', d['value'], d['ms_per_step'], d['config']['avg_xent_per_frame'])"
ASLP_BN_DIFF_PLANES=0 python3 bench.py --steps 300 --warmup 50 --headline-only | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 bn convert', d['value'], d['ms_per_step'], d['config']['avg_xent_per_frame'])"
done
} > $O/run20.log 2>&1
tail -30 $O/run20.log
