cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export CDX_UNET2_MIN_BATCH=1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "unet2 or test_fused_sample_matches_reference_fixture or full_size_properties" 2>&1 | tail -12
timeout 300 python tools/op_profile2.py 256 > gpurun_out/op2_b256.txt 2>&1; cat gpurun_out/op2_b256.txt
for B in 256 512 1024 3200; do BENCH_BATCH=$B timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=$B T=1', round(d['value']), 'traj/s', 'kernel_ms', round(d['roofline']['kernel_ms'],3), 'frac', round(d['roofline']['frac'],4))"; done
for B in 512 1024 3200; do CDX_UNET2_T=2 BENCH_BATCH=$B timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=$B T=2', round(d['value']), 'traj/s', 'kernel_ms', round(d['roofline']['kernel_ms'],3), 'frac', round(d['roofline']['frac'],4))"; done
