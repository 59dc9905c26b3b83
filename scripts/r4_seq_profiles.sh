#!/bin/bash
# round 4: the C++ sequence driver with 1 / 3 / 6 writers (12 pairs of 4K on one GPU), and the kernel time per pair by kernel
# class of 24 pairs in flight as 4 batched contexts of 6 (rocprofv3 --stats totals / pairs)
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/${1:-r4r}
mkdir -p $OUT
cd $REPO
timeout -k 10 500 python3 scripts/seq_driver_probe.py 12 > $OUT/seq_driver_probe.txt 2>&1 || { tail $OUT/seq_driver_probe.txt; exit 1; }
grep "pairs of\|==" $OUT/seq_driver_probe.txt
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=16
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/seq_stats -- python3 $REPO/scripts/seq_workload.py --pairs 24 --batch 6 --steps 8 > $OUT/seq_stats.log 2>&1 || { tail $OUT/seq_stats.log; exit 1; }
# 24 pairs x (8 timed steps + 1 warm-up)
python3 $REPO/scripts/seq_decompose.py $OUT/seq_stats 216 > $OUT/seq_decomposition.txt 2>&1
cat $OUT/seq_decomposition.txt
tail -1 $OUT/seq_stats.log
find $OUT -name "*kernel_trace.csv" -size +20M -delete
find $OUT -name "*.db" -delete
