#!/bin/bash
# A/B of two builds of libbbme.so on a bench workload (BBME_LIB): scripts/ab_lib.sh path/to/other.so cfg3 [repeats]
other="$1"; wl="${2:-cfg3}"; n="${3:-3}"
for i in $(seq $n); do
  for k in "" "BBME_LIB=$other"; do
    env $k python3 bench.py --workload $wl --steps 40 --warmup 5 --no-cpu-baseline --in-flight 8 --in-flight-deep 24 --no-other-workloads --no-host-boundary 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']
print('%-44s %s: %.4f ms/step  %.2f Mblocks/s  search %.4f ms/launch  binding %.4f  seq %.2f deep %.2f' % ('$k' or 'in-tree', '$wl', d['ms_per_step'], d['value'], r['avg_launch_ms'], r['binding']['frac'], d['sequence']['value'], d['sequence_deep']['value']))"
  done
done
