#!/bin/bash
# round 6: GROUPED GUIDED program (denoiser's stream-bound layers grouped inside the guided launch): tests, then same-box A/B
O=gpurun_out/r6l; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "grouped_guided or guided" 2>&1 | tail -12 > $O/tests.txt; cat $O/tests.txt
{
for rep in 1 2; do
  for g in 1 0; do
    CDX_UNET2_GUIDED_GROUP=$g timeout 300 python tools/bench_configs.py cfg2g:256 cfg2g:200 cfg2g:130 2>&1 | grep -v "amdgpu.ids\|Warn" | sed "s/^/GUIDED_GROUP=$g  /"
  done
done
} > $O/guided_group_ab.txt 2>&1
cat $O/guided_group_ab.txt
