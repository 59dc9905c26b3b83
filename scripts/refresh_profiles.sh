#!/bin/bash
# Runs on the GPU box (via gpurun): everything profiles/ is made from, into gpurun_out/refresh/.
#   bash scripts/refresh_profiles.sh
# Afterwards, locally: scripts/pmc_report.py and scripts/pmc_search_report.py on the pmc directories, copy the stats csv and the bench line.
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/refresh
rm -rf $OUT && mkdir -p $OUT      # NB: clean gpurun_out/refresh locally too before merging a new run
cd /tmp && export TMPDIR=/tmp
# in the shell, not in the programs: under rocprofv3 the HIP runtime is up before a program's own os.environ.setdefault runs,
# so traced and untraced runs would otherwise see different numbers of hardware queues
export GPU_MAX_HW_QUEUES=16
timeout -k 10 600 python3 $REPO/bench.py > $OUT/bench.json 2> $OUT/bench.err || exit 1
echo "bench done"
QUIET="--no-cpu-baseline --in-flight 0 --no-other-workloads"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/bench.py $QUIET > $OUT/stats.log 2>&1 || exit 1
echo "stats done"
# the same with the speculative overlap off: every k_search_fast launch is then a plain whole-level search, the kernel the
# bench line's roofline object prices (its eager HIP-event pass never speculates)
export BBME_SPECULATE=0
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_plain -- python3 $REPO/bench.py $QUIET > $OUT/stats_plain.log 2>&1 || exit 1
unset BBME_SPECULATE
echo "stats (plain) done"
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/scripts/pmc_workload.py > $OUT/pmc_fetch.log 2>&1 || exit 1
echo "pmc fetch done"
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $REPO/scripts/pmc_workload.py > $OUT/pmc_write.log 2>&1 || exit 1
echo "pmc write done"
# SQ counters of the plain search launches (VALU busy, instruction mix, LDS conflicts, waits), one group per run
bash $REPO/scripts/pmc_search.sh $OUT/pmc_sq
find $OUT -name "*kernel_trace.csv" -size +20M -delete
# traces for profiles/rNN_sweep_table.txt (speculation off: the sweeps as they are on their own) and rNN_speculation_timeline.txt
export BBME_SPECULATE=0
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_plain -- python3 $REPO/bench.py --steps 8 --warmup 2 $QUIET --profile-iters 1 --no-host-boundary > $OUT/trace_plain.log 2>&1 || exit 1
unset BBME_SPECULATE
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_spec -- python3 $REPO/bench.py --steps 8 --warmup 2 $QUIET --profile-iters 1 --no-host-boundary > $OUT/trace_spec.log 2>&1 || exit 1
python3 $REPO/scripts/trace_table.py $OUT/trace_plain > $OUT/sweep_table.txt 2>&1
python3 $REPO/scripts/spec_timeline.py $OUT/trace_spec > $OUT/spec_timeline.txt 2>&1
echo "traces done"
# the sequence leg's timeline (NB: kernel tracing itself serialises launches of different streams; see profiles/README.md)
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_seq -- python3 $REPO/scripts/seq_workload.py --pairs 8 --batch 2 --steps 4 > $OUT/trace_seq.log 2>&1 || exit 1
python3 $REPO/scripts/seq_timeline.py $OUT/trace_seq 100 > $OUT/seq_timeline.txt 2>&1 || exit 1
timeout -k 10 300 python3 $REPO/scripts/seq_scan.py --pairs 1,8 > $OUT/seq_scan.txt 2>&1 || exit 1
# the stream-concurrency probe (profiles/rNN_stream_concurrency_probe.txt): built here, the binary is not in the tree
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o $OUT/conc_probe $REPO/scripts/probes/conc_probe.hip > $OUT/conc_probe_build.log 2>&1 || { echo "conc_probe does not build"; exit 1; }
(timeout -k 5 100 $OUT/conc_probe 50 40 1 && timeout -k 5 100 $OUT/conc_probe 5 200 128) > $OUT/conc_probe.txt 2>&1 || exit 1
rm -f $OUT/conc_probe
find $OUT -name "*kernel_trace.csv" -size +20M -delete
find $OUT -name "*.db" -delete
echo "all done"
