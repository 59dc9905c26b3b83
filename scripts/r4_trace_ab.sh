#!/bin/bash
# per-sweep kernel durations (scripts/trace_table.py) of the plain graph for a list of environment settings:
#   scripts/r4_trace_ab.sh OUTDIR "BBME_MEMO=0" "BBME_MEMO=1 BBME_MEMO_FORWARD=0" ...
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=16
export BBME_SPECULATE=0
for cfg in "$@"; do
  tag=$(echo $cfg | tr ' =' '__')
  for kv in $cfg; do export $kv; done
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_$tag -- python3 $REPO/bench.py --steps 8 --warmup 2 --no-cpu-baseline --profile-iters 1 --in-flight 0 --no-host-boundary --no-other-workloads > $OUT/trace_$tag.log 2>&1 || { tail $OUT/trace_$tag.log; exit 1; }
  python3 $REPO/scripts/trace_table.py $OUT/trace_$tag > $OUT/sweep_table_$tag.txt 2>&1
  echo "== $cfg"; cat $OUT/sweep_table_$tag.txt
  for kv in $cfg; do unset ${kv%%=*}; done
  find $OUT -name "*kernel_trace.csv" -size +20M -delete
  find $OUT -name "*.db" -delete
done
