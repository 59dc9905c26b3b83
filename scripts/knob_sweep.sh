#!/bin/bash
# Development aid: the bench's `value` (ms per cfg3 step, graph replays) under scheduling knobs given as "NAME=VALUE ..." lines on stdin.
#   printf 'BBME_SPEC_WGS_PER_CU=6\nBBME_SPEC_WGS_PER_CU=10\n' | bash scripts/knob_sweep.sh OUTFILE [workload]
OUT=$1; WL=${2:-cfg3}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $REPO
run() {
  env "$@" timeout -k 10 120 python3 bench.py --workload $WL --steps 40 --warmup 5 --no-cpu-baseline --no-other-workloads --no-host-boundary --in-flight 0 --profile-iters 2 2>/dev/null |
    python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms/step  %.2f Mblocks/s  search %.3f reg %.3f' % (d['ms_per_step'], d['value'], d['device_ms']['search_ms'], d['device_ms']['regularize_ms']))"
}
echo "baseline: $(run BBME_DUMMY=1)" | tee -a $OUT
echo "baseline: $(run BBME_DUMMY=1)" | tee -a $OUT
while read -r line; do
  [ -z "$line" ] && continue
  echo "$line: $(run $line)" | tee -a $OUT
done
