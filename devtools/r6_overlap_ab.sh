#!/bin/bash
# Round 6, VERDICT r5 #6: which of cfg2's six weight-gradient products belong on the side stream?  One box, interleaved rounds of
# bench.py's timed cfg2 step (500 steps behind 50 warm-up: clocks settled) with ASLP_UPDATES_ON_MAIN=n -- the n LOWEST AffineTransforms
# keep their update on the main stream, the others go to the side stream -- and with the side stream off; then the kernel timeline of one
# steady-state step for the shipped setting, the side stream off and the two-widest-layers-only setting.
#   devtools/r6_overlap_ab.sh [rounds]     -> gpurun_out/r6/overlap_ab.txt, gpurun_out/r6/cfg2_timeline_{default,serial,top2}.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6
N=${1:-3}
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
one() {   # one "label" "ENV=.. ENV=.." "flags"
  v=$(env $2 python3 $R/bench.py --headline-only --no-gemm-profile $3 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f %d' % (d['ms_per_step'], d['value']))")
  echo "$1 $v"
}
{
  echo "# bench.py --headline-only --no-gemm-profile (500 timed steps), ms per step and frames/s; $N interleaved rounds on one box"
  for i in $(seq $N); do
    one "round$i shipped(on_main=1)" "X=0" ""
    one "round$i serial(no-side-stream)" "X=0" "--no-update-overlap"
    for n in 0 2 3 4 5; do one "round$i on_main=$n" "ASLP_UPDATES_ON_MAIN=$n" ""; done
  done
} > $O/overlap_ab.txt 2>&1
cat $O/overlap_ab.txt
tl() {   # tl tag "ENV" "flags"
  rm -rf /tmp/c2_tl
  env $2 rocprofv3 --kernel-trace -d /tmp/c2_tl -o c2 -- python3 $R/bench.py --steps 20 --warmup 10 --headline-only $3 > /tmp/c2_tl.log 2>&1
  python3 $R/devtools/prof_timeline.py $(find /tmp/c2_tl -name "*.db" | head -1) xent_rows_kernel 12 > $O/cfg2_timeline_$1.txt 2>&1
  echo "$1: $(tail -1 $O/cfg2_timeline_$1.txt)"
}
tl default "X=0" ""
tl serial "X=0" "--no-update-overlap"
tl top2 "ASLP_UPDATES_ON_MAIN=4" ""
