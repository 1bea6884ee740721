#!/bin/bash
# Q1 kernel variants on one box (run through gpurun from the repo root): the rows-r*TB+tid form against the consecutive-rows
# form (QHIP_AGG_CONS) with its sub-batch size, pipelining and rows per lane.  usage: tools/q1_cons_sweep.sh <out file> [workload]
OUT=$1; WL=${2:-q1_full}
run() {
  echo "== $*" >> $OUT
  env "$@" timeout -k 10 200 python bench.py --workload $WL --steps 10 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=l['roofline']
print('   kernel', r['kernel'], 'kernel_ms %.4f' % r['kernel_ms'], 'frac %.3f' % r['frac'], 'step ms %.4f' % l['ms_per_step'])" >> $OUT 2>&1
}
run QHIP_AGG_CONS=0
run QHIP_AGG_CONS=1 QHIP_AGG_CONS_SB=1 QHIP_AGG_CONS_PIPE=0
run QHIP_AGG_CONS=1 QHIP_AGG_CONS_SB=1 QHIP_AGG_CONS_PIPE=1
run QHIP_AGG_CONS=1 QHIP_AGG_CONS_SB=2 QHIP_AGG_CONS_PIPE=0
run QHIP_AGG_CONS=1 QHIP_AGG_CONS_SB=4 QHIP_AGG_CONS_PIPE=0
run QHIP_AGG_CONS=1 QHIP_AGG_CONS_R=2 QHIP_AGG_CONS_SB=2 QHIP_AGG_CONS_PIPE=1
run QHIP_AGG_CONS=1 QHIP_AGG_CONS_R=8 QHIP_AGG_CONS_SB=1 QHIP_AGG_CONS_PIPE=0
run QHIP_AGG_CONS=1 QHIP_AGG_CONS_SB=1 QHIP_AGG_CONS_PIPE=0 QHIP_AGG_BLOCKS_PER_CU=3
run QHIP_AGG_CONS=1 QHIP_AGG_CONS_SB=1 QHIP_AGG_CONS_PIPE=1 QHIP_AGG_BLOCKS_PER_CU=2
cat $OUT
