cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3n
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r3n/gputests.log 2>&1; tail -6 gpurun_out/r3n/gputests.log
for v in 0 1 0 1; do echo "GN_VEC=$v"; CDX_GN_VEC=$v timeout 300 python tools/bench_configs.py cfg3 2>/dev/null | tail -1 | cut -c1-250; done | tee gpurun_out/r3n/ab_gnvec.txt
