#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
{
echo "== default 256"; python3 devtools/bench_s16_shapes256.py 2>&1 | grep -v Exception -A0 | grep "x"
echo "== cfg 305"; ASLP_GEMM_SPLIT_F16_TILE=305 python3 devtools/bench_s16_shapes256.py 2>&1 | grep " x "
python3 -m pytest tests/test_gemm_split16_gpu.py tests/test_fullsize_gpu.py tests/test_nnet_gpu.py -x -q -m gpu 2>&1 | tail -3
python3 bench.py --steps 300 --warmup 50 --headline-only | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2', d['value'], d['ms_per_step'])"
python3 devtools/bench_cfg1.py 2>&1 | tail -1
ASLP_GEMM_S16_SMALL=0 python3 devtools/bench_cfg1.py 2>&1 | tail -1
python3 devtools/bench_cfg1.py 2>&1 | tail -1
python3 devtools/bench_lc.py 32 30 2>&1 | grep ms/step
} > $O/run18.log 2>&1
tail -40 $O/run18.log
