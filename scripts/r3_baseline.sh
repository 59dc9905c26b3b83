#!/bin/bash
# round-3 baseline on the GPU box: tests, bench line, search PMC groups, sweep-table trace, 8-pairs-in-flight trace
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/${1:-r3a}
mkdir -p $OUT
cd $REPO
timeout -k 10 300 python3 -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1 || { tail -20 $OUT/pytest.log; exit 1; }
echo "tests done"; tail -2 $OUT/pytest.log
timeout -k 10 300 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
echo "bench done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters.txt 2>&1
bash $REPO/scripts/pmc_search.sh $OUT/pmc
export BBME_SPECULATE=0
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_plain -- python3 $REPO/bench.py --steps 8 --warmup 2 --no-cpu-baseline --profile-iters 1 --in-flight 0 --no-host-boundary > $OUT/trace_plain.log 2>&1 || exit 1
unset BBME_SPECULATE
python3 $REPO/scripts/trace_table.py $OUT/trace_plain > $OUT/sweep_table.txt 2>&1
echo "trace done"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_seq -- python3 $REPO/scripts/seq_workload.py --pairs 8 --steps 4 > $OUT/trace_seq.log 2>&1 || exit 1
python3 $REPO/scripts/seq_timeline.py $OUT/trace_seq 100 > $OUT/seq_timeline.txt 2>&1
echo "seq trace done"
find $OUT -name "*kernel_trace.csv" -size +20M -delete
find $OUT -name "*.db" -delete
