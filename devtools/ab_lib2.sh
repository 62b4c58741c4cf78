#!/bin/bash
# A/B of builds of libaslp_hip.so on the headline step, alternating: devtools/ab_lib2.sh [rounds] name=path ...   ("new" = the tree's)
rounds=${1:-3}; shift
cp kaldi-aslp_amd/libaslp_hip.so /tmp/new.so
[ $# -eq 0 ] && set -- base=devtools/_ab/base_hip.so new=/tmp/new.so
for i in $(seq $rounds); do
  for nv in "$@"; do
    v=${nv%%=*}; f=${nv#*=}
    cp $f kaldi-aslp_amd/libaslp_hip.so
    python bench.py --headline-only --steps 1000 --warmup 100 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$v', round(d['value']), round(d['ms_per_step'], 4))"
  done
done
cp /tmp/new.so kaldi-aslp_amd/libaslp_hip.so
