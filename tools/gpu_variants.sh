# A/B of compile-time kernel variants: one libcdx build per variant under build_variants/ (git-ignored), selected with CDX_LIB.
cd $GRAFT_REPO_ROOT
for v in $VARIANTS; do
  for cfg in "256 1" "512 2"; do
    set -- $cfg
    CDX_LIB=$GRAFT_REPO_ROOT/build_variants/libcdx_$v.so BENCH_BATCH=$1 CDX_UNET2_T=$2 timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v B=$1 T=$2', round(d['value']), 'traj/s', 'kernel_ms', round(d['roofline']['kernel_ms'],3), 'frac', round(d['roofline']['frac'],4))"
  done
done
CDX_LIB=$GRAFT_REPO_ROOT/build_variants/libcdx_$CHECK.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "unet2 or guided or test_fused_sample_matches_reference_fixture" 2>&1 | tail -2
