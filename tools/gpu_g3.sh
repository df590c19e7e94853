cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "guided" 2>&1 | tail -4
for B in 600; do
  timeout 300 python tools/bench_configs.py cfg2g:$B 2>&1 | tail -1 | cut -c1-200
  CDX_UNET2_T3=0 timeout 300 python tools/bench_configs.py cfg2g:$B 2>&1 | tail -1 | cut -c1-200
done
