cd $GRAFT_REPO_ROOT
for TUNE in 0 2; do
for cfg in "256 x" "512 x" "768 x" "3200 x"; do
  set -- $cfg
  CDX_UNET2_TUNE=$TUNE BENCH_BATCH=$1 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('TUNE=$TUNE B=$1', round(d['value']), 'traj/s', 'kernel_ms', round(d['roofline']['kernel_ms'],3), 'frac', round(d['roofline']['frac'],4))"
done
done
