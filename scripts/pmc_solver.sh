#!/bin/bash
# Development aid: rocprofv3 --pmc passes (one counter group per run, kernel trace only) over an eager
# pyramid, to see how the solver's memory operations behave.  Usage (on the GPU box):
#   bash scripts/pmc_solver.sh OUTDIR
OUT=$1
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
export BBME_NO_GRAPH=1
i=0
for group in \
  "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" \
  "TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum" \
  "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_PROBE_sum TCC_TAG_STALL_sum" \
  "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_PENDING_STALL_CYCLES_sum" \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_INSTS_VALU SQ_INSTS_SALU" \
  "TCC_CC_REQ_sum TCC_UC_REQ_sum TCC_RW_REQ_sum TCC_NC_REQ_sum"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $group --kernel-trace --output-format csv -d $OUT/g$i -- python3 $REPO/scripts/pmc_workload.py --iters 1 --calib-mib 64 > $OUT.g$i.log 2>&1 && echo "group $i done" || echo "group $i FAILED"
done
