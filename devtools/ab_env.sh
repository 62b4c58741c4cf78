#!/bin/bash
# A/B of an environment switch on ONE box: devtools/ab_env.sh VAR "cmd ..." [rounds]   (runs cmd with VAR=1, VAR=0 alternately)
VAR=$1; CMD=$2; N=${3:-3}
cd /tmp
for i in $(seq $N); do
  for v in 1 0; do echo "== $VAR=$v"; env $VAR=$v bash -c "$CMD" 2>&1 | tail -2; done
done
