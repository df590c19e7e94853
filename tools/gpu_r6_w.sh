#!/bin/bash
# round 6: which ops of the small-batch (split) program are cut (CDX_UNET2_SPLIT_MIN records per wave) -- sweep at B = 32 / 64 / 128, unguided + guided B = 32
O=gpurun_out/r6w; mkdir -p $O
{
for m in 32 0 4 8 12 32 8; do
  echo "== CDX_UNET2_SPLIT_MIN=$m"
  CDX_UNET2_SPLIT_MIN=$m timeout 300 python tools/time_cfg2.py 32 64 128 2>&1 | grep -v amdgpu.ids | cut -c1-140
  CDX_UNET2_SPLIT_MIN=$m timeout 300 python tools/bench_configs.py cfg2g:32 2>&1 | grep -v "amdgpu.ids\|Warn" | cut -c95-200
done
} > $O/split_min.txt 2>&1
cat $O/split_min.txt
