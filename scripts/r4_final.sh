#!/bin/bash
# round-4 end state on the GPU box: tests, the bench line, the speculative timeline, the sequence decomposition, the C++ sequence driver
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/${1:-r4final}
mkdir -p $OUT
cd $REPO
export GPU_MAX_HW_QUEUES=16
timeout -k 10 500 python3 -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log; exit 1; }
echo "tests done"; tail -2 $OUT/pytest.log
timeout -k 10 600 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail $OUT/bench.err; exit 1; }
echo "bench done"
timeout -k 10 300 python3 bench.py --force-dist --steps 20 --warmup 3 --no-cpu-baseline --no-other-workloads --no-host-boundary --in-flight 0 > $OUT/bench_force_dist.json 2> $OUT/bench_force_dist.err || { tail $OUT/bench_force_dist.err; exit 1; }
echo "force-dist done"
cd /tmp && export TMPDIR=/tmp
QUIET="--no-cpu-baseline --in-flight 0 --no-other-workloads"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/bench.py $QUIET > $OUT/stats.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_spec -- python3 $REPO/bench.py --steps 8 --warmup 2 $QUIET --profile-iters 1 --no-host-boundary > $OUT/trace_spec.log 2>&1 || exit 1
python3 $REPO/scripts/spec_timeline.py $OUT/trace_spec > $OUT/spec_timeline.txt 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/seq_stats -- python3 $REPO/scripts/seq_workload.py --pairs 24 --batch 6 --steps 8 > $OUT/seq_stats.log 2>&1 || { tail $OUT/seq_stats.log; exit 1; }
python3 $REPO/scripts/seq_decompose.py $OUT/seq_stats 216 > $OUT/seq_decomposition.txt 2>&1
echo "traces done"
cd $REPO
timeout -k 10 500 python3 scripts/seq_driver_probe.py 12 > $OUT/seq_driver_probe.txt 2>&1 || { tail $OUT/seq_driver_probe.txt; exit 1; }
grep "pairs of\|==" $OUT/seq_driver_probe.txt
find $OUT -name "*kernel_trace.csv" -size +20M -delete
find $OUT -name "*.db" -delete
echo "all done"
