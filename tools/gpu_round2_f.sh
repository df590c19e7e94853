cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "guided or unet2 or split_tail or steady_state or full_size or test_fused_sample_matches_reference_fixture or shipped" 2>&1 | tail -4
for FS in 1 0; do
for cfg in "256 1" "512 2" "3200 2"; do
  set -- $cfg
  CDX_UNET2_FUSE_SKIP=$FS BENCH_BATCH=$1 CDX_UNET2_T=$2 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('FUSE=$FS B=$1 T=$2', round(d['value']), 'traj/s', 'kernel_ms', round(d['roofline']['kernel_ms'],3), 'frac', round(d['roofline']['frac'],4))"
done
for B in 256 3200; do CDX_UNET2_FUSE_SKIP=$FS timeout 300 python tools/bench_configs.py cfg2g:$B 2>/dev/null | tail -1 | cut -c1-200; done
done
timeout 300 python tools/op_profile2.py 256 1 8 > gpurun_out/op2_b256_w8.txt 2>&1; tail -1 gpurun_out/op2_b256_w8.txt
