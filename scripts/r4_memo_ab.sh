#!/bin/bash
# round-4 A/B on the GPU box: GPU tests with the SAD memo, then the per-sweep solver counters and the bench line with the memo
# off / on without forwarding / on with forwarding
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/${1:-r4b}
mkdir -p $OUT
cd $REPO
timeout -k 10 400 python3 -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log; exit 1; }
echo "tests done"; tail -2 $OUT/pytest.log
for cfg in "BBME_MEMO=0" "BBME_MEMO=1 BBME_MEMO_FORWARD=0" "BBME_MEMO=1 BBME_MEMO_FORWARD=1"; do
  tag=$(echo $cfg | tr ' =' '__')
  env $cfg timeout -k 10 200 python3 scripts/sweep_timeline.py > $OUT/sweeps_$tag.txt 2>&1 || { tail $OUT/sweeps_$tag.txt; exit 1; }
  env $cfg timeout -k 10 200 python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --in-flight 8 --in-flight-deep 24 --no-other-workloads --no-host-boundary > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err || { tail $OUT/bench_$tag.err; exit 1; }
  python3 - <<PY
import json
d=json.loads(open("$OUT/bench_$tag.json").readline())
print("%-40s %.4f ms/step %.2f Mblocks/s  reg %.4f search %.4f  seq %.2f deep %.2f" % ("$cfg", d["ms_per_step"], d["value"], d["regularizer"]["ms"], d["roofline"]["avg_launch_ms"], d["sequence"]["value"], d["sequence_deep"]["value"]))
PY
done
