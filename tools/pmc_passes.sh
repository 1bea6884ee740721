#!/bin/bash
# PMC passes over one bench.py workload (run on the GPU box through gpurun, from the repo root):
#   tools/pmc_passes.sh <out_dir> <tag> <bench args...>
# One rocprofv3 run per counter group (separate --pmc passes, with --kernel-trace only: MI355X_MICROARCH.md
# "rocprofv3 PMC slots": TCC has 4 slots, FETCH_SIZE costs 3, WRITE_SIZE 2, SQ 8), then one --kernel-trace --stats run.
# tools/summarize_pmc.py turns the CSVs into profiles/rNN_*.json.
set -o pipefail
export TMPDIR=/tmp
OUT=$1; TAG=$2; shift 2
mkdir -p $OUT
GROUPS_=(
  "FETCH_SIZE"
  "WRITE_SIZE"
  "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"
  "TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum TCC_ATOMIC_sum TCC_TAG_STALL_sum"
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM"
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM"
  "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_UTCL1_TRANSLATION_MISS_sum"
  "TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_UTCL1_REQUEST_sum"
  "GRBM_GUI_ACTIVE GRBM_COUNT"
)
# PMC_ONLY="0 1 4": only these groups (indices above); default: all
i=0
for g in "${GROUPS_[@]}"; do
  if [ -n "$PMC_ONLY" ] && ! echo " $PMC_ONLY " | grep -q " $i "; then i=$((i+1)); continue; fi
  timeout -k 10 300 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $OUT/pmc_${TAG}_$i -o $TAG -- python3 bench.py "$@" > $OUT/pmc_${TAG}_$i.json 2> $OUT/pmc_${TAG}_$i.err
  echo "pmc pass $i ($g) exit $?"
  i=$((i+1))
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$TAG -o $TAG -- python3 bench.py "$@" > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err
echo "trace exit $?"
