#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $REPO
for cfg in "BBME_BENCH_NO_PAIRCHECK=1" "BBME_X=1"; do
  env $cfg timeout -k 10 300 python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --in-flight 8 --in-flight-deep 24 --no-host-boundary 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline())
print('$cfg', d['ms_per_step'], d['sequence']['value'], d['sequence_deep']['value'])
for k, v in d['other_workloads'].items(): print('  ', k, v['value'], 'seq8', v['sequence_8_pairs']['value'], 'deep', v.get('sequence_deep', {}).get('value'))"
done
