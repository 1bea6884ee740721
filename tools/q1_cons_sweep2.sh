#!/bin/bash
# (second Q1 sweep of round 4: the two-register-set loop without consecutive rows, and q1_mini; profiles/r04_q1_consecutive_rows.txt)
OUT=$1
run() {
  WL=$1; shift
  echo "== $WL $*" >> $OUT
  env "$@" timeout -k 10 200 python bench.py --workload $WL --steps 10 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=l['roofline']
print('   kernel', r['kernel'], 'kernel_ms %.4f' % r['kernel_ms'], 'frac %.3f' % r['frac'], 'step ms %.4f' % l['ms_per_step'])" >> $OUT 2>&1
}
run q1_full QHIP_AGG_CONS=0 QHIP_AGG_PIPE=1
run q1_full QHIP_AGG_CONS=0 QHIP_AGG_PIPE=1 QHIP_AGG_R=1
run q1_full QHIP_AGG_CONS=0 QHIP_AGG_PIPE=1 QHIP_AGG_R=3
run q1_full QHIP_AGG_CONS=1 QHIP_AGG_CONS_SB=4 QHIP_AGG_CONS_PIPE=0 QHIP_AGG_BLOCKS_PER_CU=3
run q1_full QHIP_AGG_CONS=1 QHIP_AGG_CONS_R=2 QHIP_AGG_CONS_SB=2 QHIP_AGG_CONS_PIPE=0
run q1_mini QHIP_AGG_CONS=0
run q1_mini QHIP_AGG_CONS=1 QHIP_AGG_CONS_SB=4 QHIP_AGG_CONS_PIPE=0
run q1_mini QHIP_AGG_CONS=1 QHIP_AGG_CONS_SB=4 QHIP_AGG_CONS_PIPE=1
run q1_mini QHIP_AGG_CONS=1 QHIP_AGG_CONS_SB=1 QHIP_AGG_CONS_PIPE=1
run q1_mini QHIP_AGG_CONS=0 QHIP_AGG_PIPE=1
cat $OUT
