for k in "X=0" "BBME_BENCH_WRITERS=6" "BBME_BENCH_WRITERS=6 BBME_WRITER_THREADS=4" "BBME_BENCH_WRITERS=4 BBME_WRITER_THREADS=4" "BBME_BENCH_WRITERS=8 BBME_WRITER_THREADS=2" "BBME_BENCH_WRITERS=3 BBME_WRITER_THREADS=5"; do
  env $k python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --in-flight 0 --no-other-workloads 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); h=d['host_boundary']
print('%-50s cells e2e %.2f ms  alone %.2f  async dense %.2f  to_cells %.2f' % ('$k', h['end_to_end_cells_writer_ms'], h['cells_writer_alone_ms'], h['end_to_end_async_writer_ms'], h['host_frames_to_cells_ms']))"
done
