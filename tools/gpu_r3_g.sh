cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3g
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r3g/gputests.log 2>&1; tail -8 gpurun_out/r3g/gputests.log
for B in 256 512 32 64 128; do
  BENCH_BATCH=$B timeout 300 python bench.py --steps 80 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=$B', round(d['value']), 'traj/s', 'ms_per_call', round(d['ms_per_step'],3), 'kernel_ms', round(d['roofline']['kernel_ms'],3), 'frac', round(d['roofline']['frac'],4))"
done | tee gpurun_out/r3g/batch_sweep_small.txt
