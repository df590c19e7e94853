# rocprofv3 kernel statistics of the guided configurations (one program-kernel launch per guided sample() call)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/gs
cd /tmp && export TMPDIR=/tmp
for cfg in cfg2g:256 cfg2g:3200 cfgKg:256 cfgAg:256; do
  tag=$(echo $cfg | tr ':' '_')
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/gs/$tag -- python $R/tools/bench_configs.py $cfg > $R/gpurun_out/gs/$tag.log 2>&1
  f=$(find $R/gpurun_out/gs/$tag -name "*kernel_stats.csv" | head -1)
  cp "$f" $R/gpurun_out/gs/r02_${tag}_rocprofv3_kernel_stats.csv
  head -4 $R/gpurun_out/gs/r02_${tag}_rocprofv3_kernel_stats.csv | cut -c1-170
  rm -rf $R/gpurun_out/gs/$tag
done
