#!/bin/bash
# round 6: SMALL-BATCH guided program (one trajectory over 2 / 4 workgroups, denoiser ops cut): tests, then same-box A/B
O=gpurun_out/r6n; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "guided or granule" 2>&1 | tail -4 > $O/tests.txt; cat $O/tests.txt
{
for rep in 1 2; do
  for g in 1 0; do
    CDX_UNET2_GUIDED_SPLIT=$g timeout 300 python tools/bench_configs.py cfg2g:8 cfg2g:32 cfg2g:64 cfg2g:128 2>&1 | grep -v "amdgpu.ids\|Warn" | sed "s/^/GUIDED_SPLIT=$g  /"
  done
done
} > $O/guided_split_ab.txt 2>&1
cat $O/guided_split_ab.txt | cut -c1-200
