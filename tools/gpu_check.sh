# quick GPU regression + sweep of the north-star kernel (default library build)
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "unet2 or three_traj or split_tail or steady_state or full_size or guided or test_fused_sample_matches_reference_fixture" 2>&1 | tail -3
for B in 256 512 768 3200; do
  BENCH_BATCH=$B timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=$B', round(d['value']), 'traj/s', 'kernel_ms', round(d['roofline']['kernel_ms'],3), 'frac', round(d['roofline']['frac'],4))"
done
for B in 256 3200; do timeout 300 python tools/bench_configs.py cfg2g:$B 2>/dev/null | tail -1 | cut -c1-200; done
timeout 300 python tools/op_profile2.py 256 1 8 2>&1 | tail -1
