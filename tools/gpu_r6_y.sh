#!/bin/bash
# round 6: GEMM launcher thresholds re-swept on the final build: configs 3 / 4 (512 shard) / 5 / ChiTransformer, same box, alternating
O=gpurun_out/r6y; mkdir -p $O
run() { env "$@" timeout 600 python tools/bench_configs.py cfg3 cfg4:512 cfgT:1024:10 cfg5:16384 2>&1 | grep -v "amdgpu.ids\|Warn" | sed 's/.*"config": "\([^:,]*\).*"frac_fp32_mfma_peak": \([0-9.]*\).*/\1 \2/' | tr '\n' ' '; echo; }
{
for rep in 1 2 3; do
echo "default: $(run X=1)"
echo "XCD_ORDER=1: $(run CDX_GEMM_XCD_ORDER=1)"
echo "SMALL_TILE_BELOW=256: $(run CDX_GEMM_SMALL_TILE_BELOW=256)"
echo "SMALL_TILE_BELOW=384: $(run CDX_GEMM_SMALL_TILE_BELOW=384)"
echo "XCD_ORDER=1 + SMALL_TILE_BELOW=384: $(run CDX_GEMM_XCD_ORDER=1 CDX_GEMM_SMALL_TILE_BELOW=384)"
done
} > $O/gemm_thresholds2.txt 2>&1
cat $O/gemm_thresholds2.txt
