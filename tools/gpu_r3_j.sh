cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3j
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r3j/gputests.log 2>&1; tail -8 gpurun_out/r3j/gputests.log
timeout 300 python tools/optim_bench.py 2>/dev/null | tee gpurun_out/r3j/r03_optim_bench.jsonl
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3j/stats -- python $R/tools/optim_bench.py > $R/gpurun_out/r3j/stats.log 2>&1
f=$(find $R/gpurun_out/r3j/stats -name "*kernel_stats.csv" | head -1); cp "$f" $R/gpurun_out/r3j/r03_optim_rocprofv3_kernel_stats.csv; head -8 $R/gpurun_out/r3j/r03_optim_rocprofv3_kernel_stats.csv | cut -c1-220
rm -rf $R/gpurun_out/r3j/stats
