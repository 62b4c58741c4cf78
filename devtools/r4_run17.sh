#!/bin/bash
# no private segment on the split-fp16 product kernels; the 32 x 64 tile (cfg 304) against split-K for 256-row products; 128 x 128 vs 64 x 128 at 1920 rows
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
{
echo "== default"; python3 devtools/bench_s16_shapes.py
echo "== no 128x128"; ASLP_GEMM_S16_128_MIN=100000 python3 devtools/bench_s16_shapes.py
echo "== cfg 304"; ASLP_GEMM_SPLIT_F16_TILE=304 python3 devtools/bench_s16_shapes256.py
echo "== default 256"; python3 devtools/bench_s16_shapes256.py
python3 bench.py --steps 300 --warmup 50 --headline-only | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2', d['value'], d['ms_per_step'])"
python3 devtools/bench_cfg1.py 2>&1 | tail -2
python3 -m pytest tests/test_gemm_split16_gpu.py -x -q -m gpu 2>&1 | tail -3
} > $O/run17.log 2>&1
tail -60 $O/run17.log
