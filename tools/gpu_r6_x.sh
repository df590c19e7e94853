#!/bin/bash
# round 6: program-compiler tunables re-swept at B = 256 (grouped) / 768 / 32: shortest K slice, skip-conv fusing limit, allow 4x4 mode
O=gpurun_out/r6x; mkdir -p $O
{
echo "== default"; timeout 300 python tools/time_cfg2.py 256 768 32 2>&1 | grep -v amdgpu.ids | cut -c1-130
for v in "CDX_UNET2_MIN_SLICE=2" "CDX_UNET2_MIN_SLICE=4" "CDX_UNET2_MIN_SLICE=6" "CDX_UNET2_FUSE_MAX=60" "CDX_UNET2_FUSE_MAX=200" "CDX_UNET2_FUSE_SKIP=0" "CDX_UNET2_TUNE=1" "CDX_UNET2_TUNE=2" "CDX_UNET2_TUNE=4"; do
  echo "== $v"; env $v timeout 300 python tools/time_cfg2.py 256 768 32 2>&1 | grep -v amdgpu.ids | cut -c1-130
done
echo "== default"; timeout 300 python tools/time_cfg2.py 256 768 32 2>&1 | grep -v amdgpu.ids | cut -c1-130
} > $O/compiler_tunables.txt 2>&1
cat $O/compiler_tunables.txt
