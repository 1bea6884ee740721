#!/bin/bash
# A/B on one box: Q3 at SF10 with the dense build's bitmap packed from a byte map (round 4) / set by atomics (round 3)
OUT=$1
for v in 1 0 1 0; do
  echo "== QHIP_JOIN_DENSE_BYTEMAP=$v" >> $OUT
  QHIP_JOIN_DENSE_BYTEMAP=$v timeout -k 10 200 python bench.py --workload q3 --steps 20 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=l['records']['q3']
print('   ms per query %.4f' % r['ms_per_step'], ' '.join('%s %.1f us' % (k['operator'], k['kernel_ms']*1e3) for k in r['kernels']))" >> $OUT 2>&1
done
cat $OUT
