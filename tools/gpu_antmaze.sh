# antmaze-size Diffuser (compact guided program / compact one-trajectory program): fixture tests, then old vs new path
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "shipped_large or beyond_one_workgroup or three_traj or guided or unet2 or steady or split_tail" 2>&1 | tail -8
for B in 256 3200; do
  timeout 300 python tools/bench_configs.py cfgAu:$B 2>&1 | tail -1 | cut -c1-230
  timeout 300 python tools/bench_configs.py cfgAg:$B 2>&1 | tail -1 | cut -c1-230
done
BENCH_BATCH=768 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | cut -c1-400
