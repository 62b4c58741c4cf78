#!/bin/bash
# The engine's aslp-nnet-train-frame against the REFERENCE's own main of that name built on the engine (bin_ref/, include/aslp_compat_kaldi.h):
# the same cfg2 run from archives in the page cache (randomizer 32768, minibatch 1024), each tool's own fps line, interleaved rounds.
#   devtools/r6_refmain_e2e.sh [frames] [rounds]
R=${GRAFT_REPO_ROOT:-/root/repo}
F=${1:-1024000}; N=${2:-3}
python $R/devtools/bench_tool_e2e.py /dev/shm/e2e $F > /dev/null 2>&1
run() { $R/kaldi-aslp_amd/$1/aslp-nnet-train-frame --print-args=false --learn-rate=0.00001 --minibatch-size=1024 --randomizer-size=32768 ark:/dev/shm/e2e/feats.ark ark:/dev/shm/e2e/post.ark /dev/shm/e2e/nnet.init /dev/shm/e2e/nnet.out.$1 2>&1 | grep -oE "fps[0-9.e+]+"; rm -f /dev/shm/e2e/nnet.out.$1; }
for i in $(seq $N); do echo "engine's tool        $(run bin)"; echo "reference's main     $(run bin_ref)"; done
rm -rf /dev/shm/e2e
