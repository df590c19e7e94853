cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_reference_checkpoint.py -m gpu -x -q -k "unet2 or test_fused_sample_matches_reference_fixture or full_size_properties or shipped or ema_update or checkpoint or pearce_mlp_widths or beyond_one_workgroup or config3_width" 2>&1 | tail -12
timeout 300 python tools/op_profile2.py 256 > gpurun_out/op2_b256.txt 2>&1; tail -52 gpurun_out/op2_b256.txt | cut -c1-60 | tail -50 | awk 'NR%1==0' | head -60
for B in 256 1024 3200; do BENCH_BATCH=$B timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=$B', round(d['value']), 'traj/s', 'kernel_ms', round(d['roofline']['kernel_ms'],3), 'frac', round(d['roofline']['frac'],4))"; done
BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dist path:', round(d['value']), d['scaling'], json.dumps(d.get('strong_scaling'))[:600])"
