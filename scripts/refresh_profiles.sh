#!/bin/bash
# Runs on the GPU box (via gpurun): everything profiles/ is made from, into gpurun_out/refresh/.
#   bash scripts/refresh_profiles.sh
# Afterwards, locally: scripts/pmc_report.py on the two pmc directories, copy the stats csv and the bench line.
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/refresh
rm -rf $OUT && mkdir -p $OUT      # NB: clean gpurun_out/refresh locally too before merging a new run
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 $REPO/bench.py > $OUT/bench.json 2> $OUT/bench.err || exit 1
echo "bench done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/bench.py --no-cpu-baseline --in-flight 0 > $OUT/stats.log 2>&1 || exit 1
echo "stats done"
# the same with the speculative overlap off: every k_search_fast launch is then a plain whole-level search, the kernel the
# bench line's roofline object prices (its eager HIP-event pass never speculates)
export BBME_SPECULATE=0
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_plain -- python3 $REPO/bench.py --no-cpu-baseline --in-flight 0 > $OUT/stats_plain.log 2>&1 || exit 1
unset BBME_SPECULATE
echo "stats (plain) done"
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/scripts/pmc_workload.py > $OUT/pmc_fetch.log 2>&1 || exit 1
echo "pmc fetch done"
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $REPO/scripts/pmc_workload.py > $OUT/pmc_write.log 2>&1 || exit 1
echo "pmc write done"
find $OUT -name "*kernel_trace.csv" -size +20M -delete
# traces for profiles/rNN_sweep_table.txt (speculation off: the sweeps as they are on their own) and rNN_speculation_timeline.txt
export BBME_SPECULATE=0
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_plain -- python3 $REPO/bench.py --steps 8 --warmup 2 --no-cpu-baseline --profile-iters 1 --in-flight 0 --no-host-boundary > $OUT/trace_plain.log 2>&1 || exit 1
unset BBME_SPECULATE
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_spec -- python3 $REPO/bench.py --steps 8 --warmup 2 --no-cpu-baseline --profile-iters 1 --in-flight 0 --no-host-boundary > $OUT/trace_spec.log 2>&1 || exit 1
echo "traces done"
find $OUT -name "*kernel_trace.csv" -size +20M -delete
