#!/bin/bash
# sequence legs of a small workload under a list of environment settings: scripts/r4_seq_ab.sh OUTDIR WORKLOAD "ENV=..." ...
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/$1; WL=$2; shift; shift
mkdir -p $OUT
cd $REPO
for cfg in "$@"; do
  tag=$(echo $cfg | tr ' =' '__')
  env $cfg timeout -k 10 300 python3 bench.py --workload $WL --steps 40 --warmup 5 --no-cpu-baseline --in-flight 8 --in-flight-deep 32 --no-host-boundary --no-other-workloads > $OUT/bench_${WL}_$tag.json 2> $OUT/bench_${WL}_$tag.err || { tail $OUT/bench_${WL}_$tag.err; exit 1; }
  python3 - <<PY
import json
d=json.loads(open("$OUT/bench_${WL}_$tag.json").readline())
print("%-30s %s: %.4f ms/step %.2f Mblocks/s  reg %.4f  seq %.2f (%.4f ms/pair) deep %.2f (%.4f)" % ("$cfg", "$WL", d["ms_per_step"], d["value"], d["regularizer"]["ms"], d["sequence"]["value"], d["sequence"]["ms_per_pair"], d["sequence_deep"]["value"], d["sequence_deep"]["ms_per_pair"]))
PY
done
