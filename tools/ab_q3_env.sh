#!/bin/bash
# A/B on one box: Q3 at SF10 under two settings of ONE environment switch, alternated.  usage: tools/ab_q3_env.sh <out> <VAR> <a> <b>
OUT=$1; VAR=$2; A=$3; B=$4
for v in $A $B $A $B; do
  echo "== $VAR=$v" >> $OUT
  env $VAR=$v timeout -k 10 200 python bench.py --workload q3 --steps 20 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=l['records']['q3']
print('   ms per query %.4f  with export %.4f ' % (r['ms_per_step'], r['execute_with_export_ms']), ' '.join('%s %.1f us' % (k['operator'], k['kernel_ms']*1e3) for k in r['kernels']))" >> $OUT 2>&1
done
cat $OUT
