#!/bin/bash
# round 6: new GEMM launcher defaults (128 x 128 tiles from 256 tiles on for N > 64; XCD-ordered tile walk) against the old ones (env), same box
O=gpurun_out/r6z; mkdir -p $O
run() { env "$@" timeout 600 python tools/bench_configs.py cfg3 cfg4:512 cfgT:1024:10 cfg5:16384 2>&1 | grep -v "amdgpu.ids\|Warn" | sed 's/.*"config": "\([^:,]*\).*"frac_fp32_mfma_peak": \([0-9.]*\).*/\1 \2/' | tr '\n' ' '; echo; }
{
for rep in 1 2 3; do
echo "new defaults: $(run X=1)"
echo "old (SMALL_TILE_BELOW=520 XCD_ORDER=0): $(run CDX_GEMM_SMALL_TILE_BELOW=520 CDX_GEMM_XCD_ORDER=0)"
done
export UPDATE_BENCH_GRAPH_ONLY=1
echo "new:"; timeout 600 python tools/update_bench.py cfg2 cfg3 cfg4 cfg5 chitf 2>&1 | grep "update()"
echo "old:"; CDX_GEMM_SMALL_TILE_BELOW=520 CDX_GEMM_XCD_ORDER=0 timeout 600 python tools/update_bench.py cfg2 cfg3 cfg4 cfg5 chitf 2>&1 | grep "update()"
} > $O/gemm_defaults_ab.txt 2>&1
cat $O/gemm_defaults_ab.txt
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -2
