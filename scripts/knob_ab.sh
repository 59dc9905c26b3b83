#!/bin/bash
# Development aid: interleaved repeats of the bench's `value` under a few environment settings ("NAME=VALUE ..." lines on stdin;
# an empty setting is written as "-"), median and best per setting: printf -- '-\nBBME_X=1\n' | bash scripts/knob_ab.sh [repeats] [workload]
N=${1:-3}; WL=${2:-cfg3}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $REPO
mapfile -t CFG
TMP=$(mktemp -d)
for i in $(seq $N); do
  for j in "${!CFG[@]}"; do
    c="${CFG[$j]}"; [ "$c" = "-" ] && c="BBME_DUMMY=1"
    env $c timeout -k 10 120 python3 bench.py --workload $WL --steps 60 --warmup 5 --no-cpu-baseline --no-other-workloads --no-host-boundary --in-flight 0 --profile-iters 2 2>/dev/null |
      python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['device_ms']['regularize_ms'])" >> $TMP/$j.txt
  done
done
for j in "${!CFG[@]}"; do
  python3 - "$TMP/$j.txt" "${CFG[$j]}" <<'PY'
import sys, statistics
rows = [tuple(map(float, l.split())) for l in open(sys.argv[1])]
ms = sorted(r[0] for r in rows); reg = sorted(r[1] for r in rows)
print("%-60s ms/step median %.4f best %.4f  (%.2f Mblocks/s)  eager reg median %.3f   n=%d" % (sys.argv[2], statistics.median(ms), ms[0], 32640 / statistics.median(ms) / 1e3, statistics.median(reg), len(ms)))
PY
done
