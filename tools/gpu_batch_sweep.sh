# Batch-size sweep of the north-star kernel (GPU box): kernel time and fp32-MFMA fraction over batch size.  "x" = the runtime's own
# plan (runtime2.plan_for: rounds of 256 x T workgroups); a number forces trajectories per workgroup (CDX_UNET2_T) / waves (CDX_UNET2_NW).
cd $GRAFT_REPO_ROOT
for cfg in "128 x 8" "256 x 8" "256 1 4" "512 x 8" "512 1 8" "768 x 8" "768 2 8" "1024 x 8" "1536 x 8" "3200 x 8" "3200 1 8"; do
  set -- $cfg
  if [ "$2" = "x" ]; then unset CDX_UNET2_T; else export CDX_UNET2_T=$2; fi
  BENCH_BATCH=$1 CDX_UNET2_NW=$3 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=$1 T=$2 NW=$3', round(d['value']), 'traj/s', 'kernel_ms', round(d['roofline']['kernel_ms'],3), 'frac', round(d['roofline']['frac'],4), d['roofline']['kernel'][:40])"
done
