#!/bin/bash
# round 6: slots the split-K rule fills (8-wave 128 x 128 tiles: 512 resident workgroups, was 768): config 3 and the training steps, same box
O=gpurun_out/r6ac; mkdir -p $O
frac() { grep -v "amdgpu.ids\|Warn" | sed 's/.*"frac_fp32_mfma_peak": \([0-9.]*\).*/\1/' | tr '\n' ' '; }
export UPDATE_BENCH_GRAPH_ONLY=1
{
for rep in 1 2; do
for sl in 0 768 1024 256; do
  if [ $sl = 0 ]; then unset CDX_GEMM_SPLITK_SLOTS; tag="default (512 for the 8-wave shape)"; else export CDX_GEMM_SPLITK_SLOTS=$sl; tag="CDX_GEMM_SPLITK_SLOTS=$sl"; fi
  echo "$tag: config 3 $(timeout 300 python tools/bench_configs.py cfg3 2>&1 | frac)"
  timeout 300 python tools/update_bench.py cfg3 cfg5 2>&1 | grep "update()" | sed "s/^/   /"
done
done
unset CDX_GEMM_SPLITK_SLOTS
for f in 51 75 100; do echo "CDX_GEMM_SPLITK_FILL=$f: config 3 $(CDX_GEMM_SPLITK_FILL=$f timeout 300 python tools/bench_configs.py cfg3 2>&1 | frac)"; done
} > $O/splitk_slots.txt 2>&1
cat $O/splitk_slots.txt
