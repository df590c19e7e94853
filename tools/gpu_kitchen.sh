# kitchen-size guided Diffuser: reference fixture test, then program kernel vs the per-step executor (CDX_UNET2_GUIDED=0)
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "shipped_large or guided" 2>&1 | tail -4
for B in 256 1024 3200; do
  timeout 300 python tools/bench_configs.py cfgKg:$B 2>&1 | tail -1 | cut -c1-260
  CDX_UNET2_GUIDED=0 timeout 300 python tools/bench_configs.py cfgKg:$B 2>&1 | tail -1 | cut -c1-260
done
