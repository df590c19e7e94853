#!/bin/bash
# ChiUNet1d below the crossover batch: first-generation program kernel vs the implicit-GEMM executor
mkdir -p gpurun_out/r3q
out=gpurun_out/r3q/chiunet_small_batch.txt
: > $out
for spec in cfg3:8 cfg3:32 cfg3:64 cfg3:8:32 cfg3:32:32 cfg3:64:32 cfg3:8:64 cfg3:64:64; do
  for mb in 96 1; do
    echo "== $spec CDX_UNET_GEMM_MIN_BATCH=$mb" >> $out
    CDX_UNET_GEMM_MIN_BATCH=$mb timeout 300 python tools/bench_configs.py $spec 2>&1 | grep -v amdgpu.ids | cut -c1-400 >> $out
  done
done
cat $out
