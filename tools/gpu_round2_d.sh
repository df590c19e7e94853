cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "unet2 or full_size_properties" 2>&1 | tail -3
for TUNE in 0 1; do
for cfg in "256 1 8" "512 2 8" "3200 2 8"; do
  set -- $cfg
  CDX_UNET2_TUNE=$TUNE BENCH_BATCH=$1 CDX_UNET2_T=$2 CDX_UNET2_NW=$3 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('TUNE=$TUNE B=$1 T=$2 NW=$3', round(d['value']), 'traj/s', 'kernel_ms', round(d['roofline']['kernel_ms'],3), 'frac', round(d['roofline']['frac'],4))"
done
done
for TUNE in 0 1; do
CDX_UNET2_TUNE=$TUNE timeout 300 python tools/op_profile2.py 512 2 8 > gpurun_out/op2_b512_t2_w8_tune$TUNE.txt 2>&1; tail -1 gpurun_out/op2_b512_t2_w8_tune$TUNE.txt
CDX_UNET2_TUNE=$TUNE timeout 300 python tools/op_profile2.py 256 1 8 > gpurun_out/op2_b256_w8_tune$TUNE.txt 2>&1; tail -1 gpurun_out/op2_b256_w8_tune$TUNE.txt
done
