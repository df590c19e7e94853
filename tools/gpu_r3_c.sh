cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3c
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r3c/gputests.log 2>&1; tail -30 gpurun_out/r3c/gputests.log
for t in 4 8 16; do echo "== MLP tile $t"; CDX_MLP_TILE=$t timeout 300 python tools/bench_configs.py cfg1 2>&1 | tail -1 | cut -c1-400; done | tee gpurun_out/r3c/cfg1_tiles.txt
timeout 600 python bench.py --steps 200 --no-cpu-baseline --no-other-configs > gpurun_out/r3c/bench_short.json 2> gpurun_out/r3c/bench_short.err; cut -c1-700 gpurun_out/r3c/bench_short.json
