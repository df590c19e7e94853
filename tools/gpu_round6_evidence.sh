#!/bin/bash
# Round-6 evidence run (GPU box): GPU suite, smoke, bench line (with its live PMC passes), rocprofv3 kernel statistics of the same command, PMC
# passes (one counter group per run), op profile of the grouped program, rocprofv3 statistics of configs 3 / 4 / ChiTransformer, of the GUIDED
# launch and of update().  Everything lands in gpurun_out/ev6/ and is copied into profiles/ by hand.
cd $GRAFT_REPO_ROOT
E=gpurun_out/ev6
mkdir -p $E
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -rs --tb=short -W always > $E/gputests_full.log 2>&1; grep -n "FAILED\|SKIPPED\|first report" $E/gputests_full.log | head -20; tail -4 $E/gputests_full.log > $E/r06_gputests_tail.txt; tail -2 $E/r06_gputests_tail.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids > $E/r06_smoke.txt; tail -5 $E/r06_smoke.txt
timeout 1200 python bench.py > $E/r06_bench_n1.json 2> $E/r06_bench_n1.err; head -c 1800 $E/r06_bench_n1.json; echo; tail -2 $E/r06_bench_n1.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline > $E/r06_bench_driver_command.json 2>/dev/null; head -c 600 $E/r06_bench_driver_command.json; echo
timeout 300 python tools/op_profile2.py 256 group4 > $E/r06_op_profile_group4_wg0.txt 2>&1; tail -2 $E/r06_op_profile_group4_wg0.txt
timeout 300 python tools/time_cfg2.py 256 32 64 128 3200 2>&1 | grep -v amdgpu.ids > $E/r06_batch_sweep.txt; cat $E/r06_batch_sweep.txt
BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 40 --warmup 5 --no-cpu-baseline --no-other-configs > $E/r06_bench_forced_dist.json 2> $E/r06_bench_forced_dist.err; tail -c 400 $E/r06_bench_forced_dist.json; echo
timeout 600 python tools/update_bench.py 2>&1 | grep -v "amdgpu.ids\|Synchronization debug\|_cuda_set_sync" > $E/r06_update_bench.txt; cat $E/r06_update_bench.txt
for c in cfg2 chitf; do timeout 200 python tools/update_census.py $c 2>&1 | grep -v "Warning\|amdgpu.ids"; echo; done > $E/r06_update_census.txt; head -6 $E/r06_update_census.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$E/stats -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-other-configs --no-pmc > $R/$E/stats.log 2>&1
f=$(find $R/$E/stats -name "*kernel_stats.csv" | head -1); cp "$f" $R/$E/r06_rocprofv3_kernel_stats.csv; head -4 $R/$E/r06_rocprofv3_kernel_stats.csv | cut -c1-220
rm -rf $R/$E/stats
for cfg in cfg3 cfg4:512 cfgT:1024:10 cfg2g:256; do
  tag=$(echo $cfg | tr ':' '_')
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$E/stats_$tag -- python $R/tools/bench_configs.py $cfg > $R/$E/stats_$tag.log 2>&1
  f=$(find $R/$E/stats_$tag -name "*kernel_stats.csv" | head -1); cp "$f" $R/$E/r06_${tag}_rocprofv3_kernel_stats.csv; head -4 $R/$E/r06_${tag}_rocprofv3_kernel_stats.csv | cut -c1-160
  tail -1 $R/$E/stats_$tag.log | cut -c1-400
  rm -rf $R/$E/stats_$tag
done
cat > /tmp/upd.py <<'PY'
import os, sys, torch
os.environ["CDX_TRAIN_GRAPH"] = "0"          # eager nodes: the kernels of a HIP-graph replay are the same ones, launched by the graph
sys.path.insert(0, sys.argv[1] + "/tools"); sys.path.insert(0, sys.argv[1])
import update_bench as ub
for name in ("cfg2", "cfg3", "cfg4", "cfg5", "chitf"):
    agent, x0, cond, what = ub.build(name)
    for _ in range(6):
        agent.update(x0, cond) if cond is not None else agent.update(x0)
torch.cuda.synchronize()
PY
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$E/stats_upd -- python /tmp/upd.py $R > $R/$E/stats_upd.log 2>&1
f=$(find $R/$E/stats_upd -name "*kernel_stats.csv" | head -1); cp "$f" $R/$E/r06_update_rocprofv3_kernel_stats.csv; head -10 $R/$E/r06_update_rocprofv3_kernel_stats.csv | cut -c1-160
rm -rf $R/$E/stats_upd
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $grp | tr ' ' '_')
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/$E/pmc_$tag -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs --no-pmc > $R/$E/pmc_$tag.log 2>&1
  f=$(find $R/$E/pmc_$tag -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
try:
    for row in csv.DictReader(open(sys.argv[1])):
        if 'cdx_unet2_kernel' in row.get('Kernel_Name', ''):
            acc[(row['Kernel_Name'][:90], row['Counter_Name'])].append(float(row['Counter_Value']))
except Exception as e:
    print("pmc parse failed", e)
for k, v in acc.items():
    print("PMC", k[1], "mean_per_dispatch", sum(v) / len(v), "dispatches", len(v), "kernel", k[0])
PY
  rm -rf $R/$E/pmc_$tag
done 2>&1 | tee $R/$E/r06_pmc_raw.txt
