#!/bin/bash
# A/B of environment settings on the search figures of a bench workload: scripts/ab_env_search.sh cfg3 2 "" "VAR=1" "VAR=2" ...
wl="$1"; n="$2"; shift 2
for i in $(seq $n); do
  for k in "$@"; do
    env $k python3 bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline --in-flight 0 --no-other-workloads --no-host-boundary 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']
print('%-24s %s: %.4f ms/step  search %.4f ms/launch  binding %.4f  level0 %s' % ('$k' or 'baseline', '$wl', d['ms_per_step'], r['avg_launch_ms'], r['binding']['frac'], r['binding']['level0_launch']['ms']))"
  done
done
