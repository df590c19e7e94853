#!/bin/bash
# round 6: round planner of large batches (runtime2.plan_parts): exhaustive set of rounds vs the older bulk + remainder rule, same box
O=gpurun_out/r6g; mkdir -p $O
{
for rep in 1 2; do
python tools/time_cfg2.py 3200 3200:CDX_UNET2_PLAN=bulk 1576 1576:CDX_UNET2_PLAN=bulk 4096 4096:CDX_UNET2_PLAN=bulk 2304 896 896:CDX_UNET2_T=3 2>&1 | grep -v "amdgpu.ids\|Warn"
done
for plan in dp bulk; do
  CDX_UNET2_PLAN=$plan python tools/bench_configs.py cfg2g:3200 cfg2g:1576 2>&1 | grep -v "amdgpu.ids\|Warn" | sed "s/^/plan=$plan  /"
done
} > $O/plan_ab.txt 2>&1
cat $O/plan_ab.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "parts or rounds or batch or guided" 2>&1 | tail -4
