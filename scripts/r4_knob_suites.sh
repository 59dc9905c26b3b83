#!/bin/bash
# The whole GPU suite under the round-4 scheduling knobs (every one of them must leave every result bit for bit): profiles/r04_knob_suites.txt
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/${1:-r4suites}
mkdir -p $OUT
cd $REPO
for cfg in "BBME_MEMO=0" "BBME_MEMO_MIN_B=8" "BBME_MEMO_MIN_B=8 BBME_MEMO_FORWARD=1" "BBME_MEMO_FORWARD=1" "BBME_LIST_SPLIT=0" "BBME_SOLVE_WGS=128 BBME_SPEC_WGS_PER_CU=8" "BBME_SPEC_MIN_GABS=0 BBME_SPEC_WGS_PER_CU=24" "BBME_NO_GRAPH=1" \
           "BBME_SOLVE_SHARE=0" "BBME_SOLVE_WAVES=2 BBME_WIDE_THRESHOLD=2" "BBME_SOLVE_WGS=8 BBME_WIDE_THRESHOLD=100000" "BBME_RELAX_RULE=100000,4,1,1 BBME_SPEC_WGS_PER_CU=6"; do
  echo "== $cfg"
  env $cfg timeout -k 10 500 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -2
done | tee $OUT/knob_suites.txt
