#!/bin/bash
# (the third Q1 sweep of round 4: QHIP_AGG_STEADY — a steady-state tile loop — existed for this sweep only; profiles/r04_q1_consecutive_rows.txt)
OUT=$1
run() {
  WL=$1; shift
  echo "== $WL $*" >> $OUT
  env "$@" timeout -k 10 200 python bench.py --workload $WL --steps 10 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=l['roofline']
print('   kernel', r['kernel'], 'kernel_ms %.4f' % r['kernel_ms'], 'frac %.3f' % r['frac'], 'step ms %.4f' % l['ms_per_step'])" >> $OUT 2>&1
}
run q1_full QHIP_AGG_STEADY=0
run q1_full QHIP_AGG_STEADY=1
run q1_full QHIP_AGG_STEADY=1 QHIP_AGG_WAVES=4
run q1_full QHIP_AGG_STEADY=1 QHIP_AGG_WAVES=3
run q1_full QHIP_AGG_STEADY=1 QHIP_AGG_R=1
run q1_mini QHIP_AGG_STEADY=0
run q1_mini QHIP_AGG_STEADY=1
run q1_mini QHIP_AGG_STEADY=1 QHIP_AGG_CONS=1 QHIP_AGG_CONS_SB=4 QHIP_AGG_CONS_PIPE=1
run q1_mini QHIP_AGG_STEADY=0 QHIP_AGG_CONS=1 QHIP_AGG_CONS_SB=4 QHIP_AGG_CONS_PIPE=1 QHIP_AGG_CONS_R=8
cat $OUT
