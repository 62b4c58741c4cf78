#!/bin/bash
# repeat the 2-rank shm bench to catch an intermittent failure; keeps the stderr of failing runs
mkdir -p gpurun_out/r3/n2loop
for i in $(seq 1 ${1:-12}); do
  ASLP_COMM_TRANSPORT=shm timeout 300 python bench.py --gpus 2 --steps 6 --warmup 2 > gpurun_out/r3/n2loop/o$i.json 2> gpurun_out/r3/n2loop/e$i.err
  rc=$?
  echo "run $i rc=$rc"
  if [ $rc -ne 0 ]; then grep -v ResetLstmStreams gpurun_out/r3/n2loop/e$i.err | head -40; else rm -f gpurun_out/r3/n2loop/e$i.err gpurun_out/r3/n2loop/o$i.json; fi
done
