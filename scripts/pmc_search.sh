#!/bin/bash
# rocprofv3 --pmc passes over the plain search launches (one counter group per run, kernel trace only; program directly
# after --).  Usage (on the GPU box):  bash scripts/pmc_search.sh OUTDIR   then  python scripts/pmc_search_report.py OUTDIR
OUT=$1
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export BBME_NO_GRAPH=1
i=0
for group in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" \
  "SQ_LEVEL_WAVES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $group --kernel-trace --output-format csv -d $OUT/g$i -- python3 $REPO/scripts/pmc_workload.py --iters 2 --calib-mib 64 > $OUT/g$i.log 2>&1 && echo "group $i done" || echo "group $i FAILED"
done
