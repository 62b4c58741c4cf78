#!/bin/bash
# A/B of the e2e tool run under environment switches, interleaved on one box: devtools/ab_e2e.sh "ENV_A" "ENV_B" [frames] [rounds]
R=${GRAFT_REPO_ROOT:-/root/repo}
F=${3:-1024000}; N=${4:-3}
python $R/devtools/bench_tool_e2e.py /dev/shm/e2e $F > /dev/null 2>&1
run() { env $1 $R/kaldi-aslp_amd/bin/aslp-nnet-train-frame --print-args=false --learn-rate=0.00001 --minibatch-size=1024 --randomizer-size=32768 ark:/dev/shm/e2e/feats.ark ark:/dev/shm/e2e/post.ark /dev/shm/e2e/nnet.init /dev/shm/e2e/nnet.out.$2 2>&1 | grep -oE "fps[0-9.e+]+"; rm -f /dev/shm/e2e/nnet.out.$2; }
for i in $(seq $N); do echo "A [$1] $(run "$1" a)"; echo "B [$2] $(run "$2" b)"; done
