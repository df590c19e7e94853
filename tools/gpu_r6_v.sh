#!/bin/bash
# round 6: which layers are grouped (CDX_UNET2_GROUP_MIN_KB) with the fast exchange in place -- sweep at B = 256, unguided and guided
O=gpurun_out/r6v; mkdir -p $O
{
for kb in 300 150 200 250 400 700 300; do
  echo "== CDX_UNET2_GROUP_MIN_KB=$kb"
  CDX_UNET2_GROUP_MIN_KB=$kb timeout 300 python tools/time_cfg2.py 256 2>&1 | grep -v amdgpu.ids | cut -c1-140
  CDX_UNET2_GROUP_MIN_KB=$kb timeout 300 python tools/bench_configs.py cfg2g:256 2>&1 | grep -v "amdgpu.ids\|Warn" | cut -c95-200
done
} > $O/group_min_kb.txt 2>&1
cat $O/group_min_kb.txt
