#!/bin/bash
# round-4 profile set: cfg2 bench line + kernel stats + PMC passes, cfg3 kernel stats, timelines, split-kernel PMC and product timings
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O/r4
bash $R/devtools/measure.sh r04_dnn_cfg2 > $O/r4/measure_cfg2.log 2>&1
bash $R/devtools/measure_lc.sh r04_lcblstm_cfg3 > $O/r4/measure_lc.log 2>&1
bash $R/devtools/prof_cfg2_timeline.sh > /dev/null 2>&1; cp $O/cfg2_timeline.txt $O/r04_dnn_cfg2_step_timeline.txt
bash $R/devtools/prof_lc_timeline.sh > /dev/null 2>&1; cp $O/lc_timeline.txt $O/r04_lcblstm_cfg3_step_timeline.txt 2>/dev/null
bash $R/devtools/prof_cfg1_timeline.sh > /dev/null 2>&1; cp $O/cfg1_timeline.txt $O/r04_dnn_cfg1_step_timeline.txt 2>/dev/null
cd /tmp; python3 $R/devtools/bench_split16.py > $O/r04_gemm_split16_bench.txt 2>&1
bash $R/devtools/pmc_s16.sh > $O/r4/pmc_s16_run.log 2>&1; cp $O/r4/pmc_s16.txt $O/r04_gemm_split16_pmc.txt 2>/dev/null
tail -c 2500 $O/r04_dnn_cfg2_bench.json
head -30 $O/r04_dnn_cfg2_kernel_stats.txt
cat $O/r04_lcblstm_cfg3_bench.txt
