cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/r02_gputest.log; cat gpurun_out/r02_gputest.log
timeout 900 python bench.py > gpurun_out/r02_bench_n1.json 2> gpurun_out/r02_bench_n1.err; cat gpurun_out/r02_bench_n1.json; tail -3 gpurun_out/r02_bench_n1.err
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmc_$grp -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs > $R/gpurun_out/pmc_$grp.log 2>&1
  f=$(find $R/gpurun_out/pmc_$grp -name "*counter_collection.csv" | head -1)
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
    print("PMC", k, "sum_over_rows", sum(v), "rows", len(v))
PY
done
