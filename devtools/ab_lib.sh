#!/bin/bash
# A/B of two builds of libaslp_hip.so on ONE box: devtools/ab_lib.sh "cmd" [rounds]; expects devtools/ab/libaslp_hip_{old,new}.so
CMD=$1; N=${2:-3}
R=${GRAFT_REPO_ROOT:-/root/repo}
for i in $(seq $N); do
  for v in new old; do cp $R/devtools/ab/libaslp_hip_$v.so $R/kaldi-aslp_amd/libaslp_hip.so; echo "== $v"; (cd /tmp; bash -c "$CMD" 2>&1 | tail -${TAILN:-1}); done
done
cp $R/devtools/ab/libaslp_hip_new.so $R/kaldi-aslp_amd/libaslp_hip.so
