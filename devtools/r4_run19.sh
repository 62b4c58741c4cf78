#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
{
python3 -m pytest tests/test_nnet_gpu.py tests/test_fullsize_gpu.py tests/test_tools_gpu.py tests/test_parallel_gpu.py tests/test_gemm_split16_gpu.py tests/test_bench_gpu.py -x -q -m gpu 2>&1 | tail -4
for i in 1 2; do
python3 bench.py --steps 300 --warmup 50 --headline-only | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 late join', d['value'], d['ms_per_step'], d['config']['avg_xent_per_frame'])"
ASLP_LATE_JOIN=0 python3 bench.py --steps 300 --warmup 50 --headline-only | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 join at end', d['value'], d['ms_per_step'], d['config']['avg_xent_per_frame'])"
python3 devtools/bench_cfg1.py 2>&1 | tail -1
ASLP_LATE_JOIN=0 python3 devtools/bench_cfg1.py 2>&1 | tail -1
done
} > $O/run19.log 2>&1
tail -30 $O/run19.log
