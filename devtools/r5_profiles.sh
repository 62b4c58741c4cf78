#!/bin/bash
# round-5 profile set: cfg2 bench line + kernel stats + PMC passes, cfg3 kernel stats, timelines, product timings, temporal components, soak
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O/r5
bash $R/devtools/measure.sh r05_dnn_cfg2 > $O/r5/measure_cfg2.log 2>&1
bash $R/devtools/measure_lc.sh r05_lcblstm_cfg3 > $O/r5/measure_lc.log 2>&1
bash $R/devtools/prof_cfg2_timeline.sh > /dev/null 2>&1; cp $O/cfg2_timeline.txt $O/r05_dnn_cfg2_step_timeline.txt
bash $R/devtools/prof_lc_timeline.sh > /dev/null 2>&1; cp $O/lc_timeline.txt $O/r05_lcblstm_cfg3_step_timeline.txt 2>/dev/null
cd /tmp; python3 $R/devtools/bench_split16.py > $O/r05_gemm_split16_bench.txt 2>&1
bash $R/devtools/prof_temporal.sh r05 > $O/r5/temporal.log 2>&1
# soak at the recipe's learn rate: 30,000 cfg2 steps (the kept weight planes' bound depends on lr), 10,000 LC-BLSTM steps
python3 $R/bench.py --steps 30000 --headline-only > $O/r05_dnn_cfg2_soak.json 2> $O/r5/soak.err
python3 $R/devtools/bench_lc.py 32 10000 2>&1 | grep -v "ResetLstmStreams\|amdgpu.ids" | tail -4 > $O/r05_lcblstm_cfg3_soak.txt
tail -c 1500 $O/r05_dnn_cfg2_bench.json
head -14 $O/r05_dnn_cfg2_kernel_stats.txt
cat $O/r05_lcblstm_cfg3_bench.txt
tail -c 600 $O/r05_dnn_cfg2_soak.json
cat $O/r05_lcblstm_cfg3_soak.txt
