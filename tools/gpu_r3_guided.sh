# Round-3 evidence for the classifier-guided loops: kernel statistics (ONE cdx_unet2_kernel<T, 8, true, ...> launch per call and range; the
# first program kernel's log_p launch is gone), the op profile of the guided program, and the MFMA-busy counter of the guided launches.
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/gs3
cd $R
timeout 300 python tools/op_profile2_guided.py 256 > gpurun_out/gs3/r03_op_profile_guided_h32.txt 2>&1; tail -3 gpurun_out/gs3/r03_op_profile_guided_h32.txt
cd /tmp && export TMPDIR=/tmp
for cfg in cfg2g:256 cfg2g:3200 cfgKg:256 cfgAg:256; do
  tag=$(echo $cfg | tr ':' '_')
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/gs3/$tag -- python $R/tools/bench_configs.py $cfg > $R/gpurun_out/gs3/$tag.log 2>&1
  f=$(find $R/gpurun_out/gs3/$tag -name "*kernel_stats.csv" | head -1)
  cp "$f" $R/gpurun_out/gs3/r03_${tag}_rocprofv3_kernel_stats.csv
  head -4 $R/gpurun_out/gs3/r03_${tag}_rocprofv3_kernel_stats.csv | cut -c1-170
  rm -rf $R/gpurun_out/gs3/$tag
done
for cfg in cfg2g:256 cfg2g:3200; do
  tag=$(echo $cfg | tr ':' '_')
  for grp in "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES"; do
    gt=$(echo $grp | tr ' ' '_')
    timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/gs3/pmc_${tag}_$gt -- python $R/tools/bench_configs.py $cfg > $R/gpurun_out/gs3/pmc_${tag}_$gt.log 2>&1
    f=$(find $R/gpurun_out/gs3/pmc_${tag}_$gt -name "*counter_collection.csv" | head -1)
    python - "$f" "$cfg" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
try:
    for row in csv.DictReader(open(sys.argv[1])):
        if 'cdx_unet2_kernel' in row.get('Kernel_Name', ''):
            acc[(row['Kernel_Name'][:60], row['Counter_Name'])].append(float(row['Counter_Value']))
except Exception as e:
    print("pmc parse failed", e)
for k, v in acc.items():
    print("PMC", sys.argv[2], k[0], k[1], "mean_per_dispatch", sum(v) / len(v), "dispatches", len(v))
PY
    rm -rf $R/gpurun_out/gs3/pmc_${tag}_$gt
  done
done 2>&1 | tee $R/gpurun_out/gs3/r03_guided_pmc_raw.txt
