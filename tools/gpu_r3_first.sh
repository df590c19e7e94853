# Round-3 first GPU call: whole -m gpu suite (new exact-config + loss/update fixtures), the bench line, rocprofv3 kernel statistics of
# configs 3 / 4 / 5 / ChiTransformer (round 2 only had r01 ones for these).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3a/gputests.log 2>&1; tail -15 gpurun_out/r3a/gputests.log
timeout 900 python bench.py > gpurun_out/r3a/bench_n1.json 2> gpurun_out/r3a/bench_n1.err; head -c 2500 gpurun_out/r3a/bench_n1.json; echo; tail -3 gpurun_out/r3a/bench_n1.err
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in cfg3 cfg4:512 cfg5:16384 cfgT:1024:10; do
  tag=$(echo $cfg | tr ':' '_')
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3a/stats_$tag -- python $R/tools/bench_configs.py $cfg > $R/gpurun_out/r3a/stats_$tag.log 2>&1
  f=$(find $R/gpurun_out/r3a/stats_$tag -name "*kernel_stats.csv" | head -1); cp "$f" $R/gpurun_out/r3a/r03_${tag}_rocprofv3_kernel_stats.csv
  tail -1 $R/gpurun_out/r3a/stats_$tag.log | cut -c1-400; head -5 $R/gpurun_out/r3a/r03_${tag}_rocprofv3_kernel_stats.csv | cut -c1-160
  rm -rf $R/gpurun_out/r3a/stats_$tag
done
