# Round-3 evidence run (GPU box): bench line, rocprofv3 kernel stats of the same command, PMC passes (one counter group per run),
# op profile, small-batch latency sweep, config-1 tile sweep.  Everything lands in gpurun_out/ev3/ and is copied into profiles/ by hand.
cd $GRAFT_REPO_ROOT
E=gpurun_out/ev3
mkdir -p $E
timeout 900 python bench.py > $E/r03_bench_n1.json 2> $E/r03_bench_n1.err; head -c 1500 $E/r03_bench_n1.json; echo; tail -2 $E/r03_bench_n1.err
timeout 300 python tools/op_profile2.py 256 > $E/r03_op_profile_wg0.txt 2>&1; tail -2 $E/r03_op_profile_wg0.txt
for B in 32 64 128 256 512 768 3200; do
  BENCH_BATCH=$B timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=$B', round(d['value']), 'traj/s', 'ms_per_call', round(d['ms_per_step'],3), 'kernel_ms', round(d['roofline']['kernel_ms'],3), 'frac', round(d['roofline']['frac'],4))"
done | tee $E/r03_batch_sweep.txt
for t in 4 8 16; do echo "MLP tile $t"; CDX_MLP_TILE=$t timeout 300 python tools/bench_configs.py cfg1 2>&1 | tail -1 | cut -c1-330; done | tee $E/r03_cfg1_tiles.txt
timeout 300 python tools/optim_bench.py 2>/dev/null | tee $E/r03_optim_bench.jsonl
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$E/stats -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-other-configs > $R/$E/stats.log 2>&1
f=$(find $R/$E/stats -name "*kernel_stats.csv" | head -1); cp "$f" $R/$E/r03_rocprofv3_kernel_stats.csv; head -4 $R/$E/r03_rocprofv3_kernel_stats.csv | cut -c1-200
rm -rf $R/$E/stats
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$E/stats1 -- python $R/tools/bench_configs.py cfg1 > $R/$E/stats1.log 2>&1
f=$(find $R/$E/stats1 -name "*kernel_stats.csv" | head -1); cp "$f" $R/$E/r03_cfg1_rocprofv3_kernel_stats.csv; head -3 $R/$E/r03_cfg1_rocprofv3_kernel_stats.csv | cut -c1-200
rm -rf $R/$E/stats1
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $grp | tr ' ' '_')
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/$E/pmc_$tag -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs > $R/$E/pmc_$tag.log 2>&1
  f=$(find $R/$E/pmc_$tag -name "*counter_collection.csv" | head -1)
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
  rm -rf $R/$E/pmc_$tag
done 2>&1 | tee $R/$E/r03_pmc_raw.txt
