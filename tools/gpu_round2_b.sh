cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r02_gputest.log; cat gpurun_out/r02_gputest.log
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -- python $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline > $R/gpurun_out/prof_stats.log 2>&1
find $R/gpurun_out/prof_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/r02_rocprofv3_kernel_stats.csv; head -5 $R/gpurun_out/r02_rocprofv3_kernel_stats.csv
for grp in "FETCH_SIZE WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU"; do
  tag=$(echo $grp | tr ' ' '_')
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmc_$tag -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $R/gpurun_out/pmc_$tag.log 2>&1
  f=$(find $R/gpurun_out/pmc_$tag -name "*counter_collection.csv" | head -1)
  python - "$f" "$grp" <<'PY'
import csv, sys, collections
f, grp = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
try:
    for row in csv.DictReader(open(f)):
        if 'cdx_unet2_kernel' in row.get('Kernel_Name', ''):
            acc[row['Counter_Name']].append(float(row['Counter_Value']))
except Exception as e:
    print("pmc parse failed", f, e)
# rows are per (dispatch, counter[, dimension]); sum per dispatch = total / n_dispatches
for k, v in acc.items():
    print("PMC", k, "sum_over_rows", sum(v), "rows", len(v))
PY
done
cd $R; ls gpurun_out | head -30
