#!/bin/bash
# round 6: grouped guided program -- classifier ops on the non-member run_op instantiation (default) vs the member one (-DCDX2_GUIDED_PLAIN_CLF=0)
O=gpurun_out/r6m; mkdir -p $O
{
for rep in 1 2; do
  for lib in default gclf0; do
    if [ $lib = default ]; then unset CDX_LIB; else export CDX_LIB=$PWD/build_variants/libcdx_$lib.so; fi
    timeout 300 python tools/bench_configs.py cfg2g:256 2>&1 | grep -v "amdgpu.ids\|Warn" | sed "s/^/lib=$lib  /"
  done
done
unset CDX_LIB
CDX_UNET2_GUIDED_GROUP=0 timeout 300 python tools/bench_configs.py cfg2g:256 2>&1 | grep -v "amdgpu.ids\|Warn" | sed "s/^/ordinary guided program  /"
} > $O/guided_plain_clf_ab.txt 2>&1
cat $O/guided_plain_clf_ab.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "granule or grouped_guided or guided" 2>&1 | tail -3
