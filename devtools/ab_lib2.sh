#!/bin/bash
# A/B of two builds of libaslp_hip.so on the headline step: devtools/_ab/base_hip.so against the tree's, alternating.
cp kaldi-aslp_amd/libaslp_hip.so /tmp/new.so
for i in 1 2 3; do
  for v in base new; do
    if [ $v = base ]; then cp devtools/_ab/base_hip.so kaldi-aslp_amd/libaslp_hip.so; else cp /tmp/new.so kaldi-aslp_amd/libaslp_hip.so; fi
    python bench.py --headline-only --steps 1000 --warmup 100 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$v', d['value'], d['ms_per_step'])"
  done
done
cp /tmp/new.so kaldi-aslp_amd/libaslp_hip.so
