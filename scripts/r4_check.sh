#!/bin/bash
# GPU tests + a short bench line of the in-tree library (round-4 iteration step): scripts/r4_check.sh OUTDIR [extra bench args]
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/${1:-r4x}; shift
mkdir -p $OUT
cd $REPO
timeout -k 10 500 python3 -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log; exit 1; }
echo "tests done"; tail -2 $OUT/pytest.log
timeout -k 10 300 python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --in-flight 8 --in-flight-deep 24 --no-host-boundary "$@" > $OUT/bench.json 2> $OUT/bench.err || { tail $OUT/bench.err; exit 1; }
python3 - <<PY
import json
d=json.loads(open("$OUT/bench.json").readline())
print("%.4f ms/step %.2f Mblocks/s  reg %.4f search/launch %.4f  seq %.2f deep %.2f" % (d["ms_per_step"], d["value"], d["regularizer"]["ms"], d["roofline"]["avg_launch_ms"], d["sequence"]["value"], d["sequence_deep"]["value"]))
print("step_binding", d.get("step_binding")); print("seq", d["sequence"]); print("deep", d["sequence_deep"])
for k, v in d.get("other_workloads", {}).items():
    print(k, v["value"], v["ms_per_step"], "reg", v["regularize_ms"], "seq8", v["sequence_8_pairs"]["value"], v["sequence_8_pairs"].get("all_pairs_checked"), "deep", v.get("sequence_deep", {}).get("value"), v.get("sequence_deep", {}).get("all_pairs_checked"), "parity", v["parity_vs_oracle"])
PY
