#!/bin/bash
# round 3: the whole GPU suite, then the committed measurements (bench line, rocprofv3 summaries of cfg2 and cfg3)
mkdir -p gpurun_out/r3
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/r3/final_gpu_tests.log 2>&1
echo "gpu tests rc=$?" >> gpurun_out/r3/final_gpu_tests.log
tail -5 gpurun_out/r3/final_gpu_tests.log
bash devtools/measure.sh r03_dnn_cfg2 > gpurun_out/r3/final_measure.log 2>&1
bash devtools/measure_lc.sh r03_lcblstm_cfg3 >> gpurun_out/r3/final_measure.log 2>&1
bash devtools/prof_cfg2_timeline.sh >> gpurun_out/r3/final_measure.log 2>&1 || true
bash devtools/prof_lc_timeline.sh >> gpurun_out/r3/final_measure.log 2>&1 || true
ls gpurun_out | grep r03
