#!/bin/bash
# A/B of one environment knob on a bench workload: scripts/ab_env.sh "VAR=1" cfg3 [repeats]
knob="$1"; wl="${2:-cfg3}"; n="${3:-3}"
for i in $(seq $n); do
  for k in "" "$knob"; do
    env $k python3 bench.py --workload $wl --steps 40 --warmup 5 --no-cpu-baseline --in-flight 0 --no-other-workloads --no-host-boundary 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); st=d.get('stages_eager_ms',{}) or {}
print('%-20s %s: %.4f ms/step  %.2f Mblocks/s  parity %s' % ('$k' or 'baseline', '$wl', d['ms_per_step'], d['value'], d.get('parity_vs_oracle')))"
  done
done
