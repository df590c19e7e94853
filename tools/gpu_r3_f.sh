cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3f
for v in nopipe pipe nopipe pipe; do
  for B in 256 512; do
    CDX_LIB=$GRAFT_REPO_ROOT/build_variants/libcdx_$v.so BENCH_BATCH=$B timeout 300 python bench.py --steps 80 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v B=$B', round(d['value']), 'traj/s', 'kernel_ms', round(d['roofline']['kernel_ms'],3), 'frac', round(d['roofline']['frac'],4))"
  done
done | tee gpurun_out/r3f/ab_pipe.txt
timeout 300 python tools/op_profile2.py 256 > gpurun_out/r3f/op_profile_wg0.txt 2>&1; tail -4 gpurun_out/r3f/op_profile_wg0.txt
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r3f/gputests.log 2>&1; tail -6 gpurun_out/r3f/gputests.log
