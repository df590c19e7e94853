# Batch-size sweep of the north-star kernel (GPU box): kernel time and fp32-MFMA fraction over batch size, trajectories per workgroup
# (CDX_UNET2_T) and waves per workgroup (CDX_UNET2_NW).  profiles/r02_batch_sweep.txt was produced by an earlier build with this script.
cd $GRAFT_REPO_ROOT
for cfg in "256 1 8" "256 1 4" "512 2 8" "512 1 8" "512 2 4" "1024 2 8" "1024 1 8" "3200 2 8" "3200 1 8" "3200 2 4" "128 1 8"; do
  set -- $cfg
  BENCH_BATCH=$1 CDX_UNET2_T=$2 CDX_UNET2_NW=$3 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=$1 T=$2 NW=$3', round(d['value']), 'traj/s', 'kernel_ms', round(d['roofline']['kernel_ms'],3), 'frac', round(d['roofline']['frac'],4))"
done
