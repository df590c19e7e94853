cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3i
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r3i/gputests.log 2>&1; tail -8 gpurun_out/r3i/gputests.log
for lp in 0 1; do for B in 256 3200; do echo "guided logp_in_launch=$lp B=$B"; CDX_UNET2_LOGP=$lp timeout 300 python tools/bench_configs.py cfg2g:$B 2>/dev/null | tail -1 | cut -c1-260; done; done | tee gpurun_out/r3i/ab_logp.txt
