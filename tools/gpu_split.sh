#!/bin/bash
# small-batch mode (one trajectory over k workgroups): parity, then the latency sweep with and without it
mkdir -p gpurun_out/split
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "split_program" 2>&1 | tail -30 > gpurun_out/split/tests.log
cat gpurun_out/split/tests.log
out=gpurun_out/split/small_batch_latency.txt
: > $out
for B in 32 128; do
  for sp in 0 auto; do
    CDX_UNET2_SPLIT=$sp BENCH_BATCH=$B timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=$B split=$sp', round(d['value']), 'traj/s', 'ms_per_call', round(d['ms_per_step'],3), 'kernel_ms', round(d['roofline']['kernel_ms'],3))" >> $out
  done
done
for mn in 8 16; do
  for B in 32 128; do
    CDX_UNET2_SPLIT_MIN=$mn BENCH_BATCH=$B timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=$B min_records=$mn', round(d['value']), 'traj/s', 'ms_per_call', round(d['ms_per_step'],3), 'kernel_ms', round(d['roofline']['kernel_ms'],3))" >> $out
  done
done
cat $out
