cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "unet2 or three_traj or split_tail or steady_state or full_size or guided" 2>&1 | tail -4
for cfg in "256 x" "512 x" "768 x" "768 2" "1536 x" "1536 2" "3200 x" "3200 2"; do
  set -- $cfg
  if [ "$2" = "x" ]; then unset CDX_UNET2_T; else export CDX_UNET2_T=$2; fi
  BENCH_BATCH=$1 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=$1 T=$2', round(d['value']), 'traj/s', 'kernel_ms', round(d['roofline']['kernel_ms'],3), 'frac', round(d['roofline']['frac'],4), d['roofline']['kernel'][:24])"
done
