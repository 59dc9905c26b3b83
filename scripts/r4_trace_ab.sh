#!/bin/bash
# Development aid: kernel-trace stats of the bench step (speculation off) under two settings, e.g. two builds of the library:
#   bash scripts/r4_trace_ab.sh OUTDIR "BBME_LIB=... X=1" "BBME_Y=0"
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export BBME_SPECULATE=0 GPU_MAX_HW_QUEUES=16
i=0
for setting in "$@"; do
  i=$((i+1))
  [ "$setting" = "-" ] || export $setting
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s$i -- python3 $REPO/bench.py --steps 30 --warmup 3 --no-cpu-baseline --in-flight 0 --no-other-workloads --no-host-boundary --profile-iters 1 > $OUT/s$i.log 2>&1 || exit 1
  [ "$setting" = "-" ] || for kv in $setting; do unset ${kv%%=*}; done
  find $OUT/s$i -name "*kernel_trace.csv" -delete; find $OUT/s$i -name "*.db" -delete
  echo "== $setting"; python3 - $OUT/s$i <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
tot = 0
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    if "k_reg" in r["Name"] or "k_search" in r["Name"]:
        print("%-60s calls %6s avg %8.2f us total %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
PY
done
