# A/B of compile-time kernel variants: one libcdx build per variant under build_variants/ (git-ignored), selected with CDX_LIB.
#   VARIANTS="base foo" CHECK=foo bash tools/gpu_variants.sh
cd $GRAFT_REPO_ROOT
for v in $VARIANTS; do
  for B in 256 512 768; do
    CDX_LIB=$GRAFT_REPO_ROOT/build_variants/libcdx_$v.so BENCH_BATCH=$B timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v B=$B', round(d['value']), 'traj/s', 'kernel_ms', round(d['roofline']['kernel_ms'],3), 'frac', round(d['roofline']['frac'],4))"
  done
done
CDX_LIB=$GRAFT_REPO_ROOT/build_variants/libcdx_$CHECK.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "unet2 or guided or three_traj or test_fused_sample_matches_reference_fixture" 2>&1 | tail -2
