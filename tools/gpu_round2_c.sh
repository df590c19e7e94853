cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/r02_gputest.log; cat gpurun_out/r02_gputest.log
timeout 300 python tools/op_profile2.py 256 1 8 > gpurun_out/op2_b256_w8.txt 2>&1; head -3 gpurun_out/op2_b256_w8.txt | tail -2; tail -1 gpurun_out/op2_b256_w8.txt
timeout 300 python tools/op_profile2.py 512 2 8 > gpurun_out/op2_b512_t2_w8.txt 2>&1; head -3 gpurun_out/op2_b512_t2_w8.txt | tail -2; tail -1 gpurun_out/op2_b512_t2_w8.txt
for cfg in "256 1 8" "128 1 8" "512 2 8" "1024 2 8" "3200 2 8"; do
  set -- $cfg
  BENCH_BATCH=$1 CDX_UNET2_T=$2 CDX_UNET2_NW=$3 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=$1 T=$2 NW=$3', round(d['value']), 'traj/s', 'kernel_ms', round(d['roofline']['kernel_ms'],3), 'frac', round(d['roofline']['frac'],4))"
done
