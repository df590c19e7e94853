# Round-2 evidence run (GPU box): bench line, rocprofv3 kernel stats of the same command, PMC passes (one group per run).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ev
timeout 900 python bench.py > gpurun_out/ev/r02_bench_n1.json 2> gpurun_out/ev/r02_bench_n1.err; head -c 1800 gpurun_out/ev/r02_bench_n1.json; echo; tail -3 gpurun_out/ev/r02_bench_n1.err
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ev/stats -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-other-configs > $R/gpurun_out/ev/stats.log 2>&1
f=$(find $R/gpurun_out/ev/stats -name "*kernel_stats.csv" | head -1); cp "$f" $R/gpurun_out/ev/r02_rocprofv3_kernel_stats.csv; head -4 $R/gpurun_out/ev/r02_rocprofv3_kernel_stats.csv | cut -c1-200
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU"; do
  tag=$(echo $grp | tr ' ' '_')
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/ev/pmc_$tag -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs > $R/gpurun_out/ev/pmc_$tag.log 2>&1
  f=$(find $R/gpurun_out/ev/pmc_$tag -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
try:
    for row in csv.DictReader(open(sys.argv[1])):
        if 'cdx_unet2_kernel' in row.get('Kernel_Name', ''):
            acc[row['Counter_Name']].append(float(row['Counter_Value']))
except Exception as e:
    print("pmc parse failed", e)
for k, v in acc.items():
    print("PMC", k, "mean_per_dispatch", sum(v) / len(v), "dispatches", len(v))
PY
  rm -rf $R/gpurun_out/ev/pmc_$tag
done 2>&1 | tee $R/gpurun_out/ev/pmc_summary.txt
rm -rf $R/gpurun_out/ev/stats
