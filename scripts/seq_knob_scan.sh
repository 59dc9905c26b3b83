#!/bin/bash
# Pairs in flight x pairs per batched context x a few environment knobs: ms per pair of scripts/seq_workload.py (cfg3).
run() { env "$@" python3 scripts/seq_workload.py --steps 10 --pairs $P --batch $B 2>/dev/null | sed "s/^/[$*] /"; }
for pb in "8 2" "6 2" "10 2" "12 2" "9 3" "12 3" "6 3" "4 2" "8 2"; do
  set -- $pb; P=$1; B=$2
  run X=0
done
P=8; B=2
for k in BBME_SOLVE_WGS=64 BBME_SOLVE_WGS=256 BBME_PASS1_LANES_MAX=40000 BBME_PASS1_LANES_MAX=0 BBME_RELAX_RULE=100000,4,1,1 BBME_RELAX_RULE=30000,8,1,0 BBME_SEARCH_SPLIT_BLOCKS=0 BBME_SCAN_FINE_MAX=0 BBME_WIDE_THRESHOLD=8 GPU_MAX_HW_QUEUES=8 GPU_MAX_HW_QUEUES=4; do
  run $k
done
run X=0
