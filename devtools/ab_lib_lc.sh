#!/bin/bash
# A/B of builds of libaslp_hip.so on the LC-BLSTM (cfg3) step, alternating: devtools/ab_lib_lc.sh [rounds] name=path ...   ("new" = the tree's)
rounds=${1:-3}; shift
cp kaldi-aslp_amd/libaslp_hip.so /tmp/new.so
[ $# -eq 0 ] && set -- base=devtools/_ab/base_hip.so new=/tmp/new.so
for i in $(seq $rounds); do
  for nv in "$@"; do
    v=${nv%%=*}; f=${nv#*=}
    cp $f kaldi-aslp_amd/libaslp_hip.so
    python devtools/bench_lc.py 32 300 2>&1 | grep "ms/step" | sed "s/^/$v /" | cut -c1-80
  done
done
cp /tmp/new.so kaldi-aslp_amd/libaslp_hip.so
