#!/bin/bash
# round 6: executor chunk sizes re-swept on the final build (config 4 shard 512, ChiTransformer B = 1024, config 3 B = 1024)
O=gpurun_out/r6aa; mkdir -p $O
frac() { grep -v "amdgpu.ids\|Warn" | sed 's/.*"frac_fp32_mfma_peak": \([0-9.]*\).*/\1/' | tr '\n' ' '; }
{
echo "config 4 shard: default $(timeout 300 python tools/bench_configs.py cfg4:512 2>&1 | frac)"
for c in 64 128 256 512; do echo "config 4 shard: CDX_DIT_CHUNK=$c $(CDX_DIT_CHUNK=$c timeout 300 python tools/bench_configs.py cfg4:512 2>&1 | frac)"; done
echo "ChiTransformer: default $(timeout 300 python tools/bench_configs.py cfgT:1024:10 2>&1 | frac)"
for c in 128 256 512 1024; do echo "ChiTransformer: CDX_CHITF_CHUNK=$c $(CDX_CHITF_CHUNK=$c timeout 300 python tools/bench_configs.py cfgT:1024:10 2>&1 | frac)"; done
echo "config 3: default $(timeout 300 python tools/bench_configs.py cfg3 2>&1 | frac)"
for c in 128 256 512 1024; do echo "config 3: CDX_CHIUNET_CHUNK=$c $(CDX_CHIUNET_CHUNK=$c timeout 300 python tools/bench_configs.py cfg3 2>&1 | frac)"; done
echo "config 4 shard: default $(timeout 300 python tools/bench_configs.py cfg4:512 2>&1 | frac)"
} > $O/chunks.txt 2>&1
cat $O/chunks.txt
