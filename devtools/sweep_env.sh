#!/bin/bash
# tile sweep inside bench.py (the only trustworthy A/B: clocks settle inside a 500-step run): devtools/sweep_nt.sh "ENV=.. ENV=.." ...
cd /tmp
for e in "$@"; do
  echo "== $e"; env $e python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-cfg3 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), {k:(round(v['avg_us'],1),round(v['tflops'],1)) for k,v in d['gemm_all']['variants'].items()})"
done
