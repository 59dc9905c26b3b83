#!/bin/bash
# cfg3, 24 pairs in flight as 4 batched contexts of 6 (bench.py's sequence_deep) under a few knobs; ms per pair
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $REPO
run() { env "$@" python3 scripts/seq_workload.py --steps 8 --pairs 24 --batch 6 --relax ${RELAX:-1} 2>/dev/null | sed "s/^/[$* relax=${RELAX:-1}] /"; }
run X=0
RELAX=0 run X=0
run BBME_SOLVE_WGS=64
run BBME_SOLVE_WGS=256
run BBME_MEMO=0
run BBME_PASS1_LANES_MAX=40000
run BBME_RELAX_RULE=100000,2,1,0
run BBME_WIDE_THRESHOLD=32
run BBME_SCAN_FINE_MAX=40000
run X=0
