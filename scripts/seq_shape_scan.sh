#!/bin/bash
# bench.py's `sequence` leg for several (pairs in flight, pairs per batched context), interleaved repeats.
#   scripts/seq_shape_scan.sh cfg3 3 "8 2" "12 3" ...      (environment knobs pass through)
wl="${1:-cfg3}"; n="${2:-3}"; shift 2
[ $# -eq 0 ] && set -- "8 2" "9 3" "12 3" "12 2" "6 3" "6 2" "12 4" "15 3"
for i in $(seq $n); do
  for pb in "$@"; do
    set_ () { P=$1; B=$2; }; set_ $pb
    python3 bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline --in-flight $P --seq-batch $B --no-other-workloads --no-host-boundary 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); s=d['sequence']
print('$wl in flight %2d as %d x %d: %.2f Mblocks/s  %.4f ms/pair   (single pair %.2f)' % (s['pairs_in_flight'], s['contexts'], s['pairs_per_context'], s['value'], s['ms_per_pair'], d['value']))"
  done
done
