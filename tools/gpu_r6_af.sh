#!/bin/bash
# round 6: max-ILP scheduling for the other translation units (cdx_train / cdx_bigbatch / cdx_optim), same box
O=gpurun_out/r6af; mkdir -p $O
export UPDATE_BENCH_GRAPH_ONLY=1
frac() { grep -v "amdgpu.ids\|Warn" | sed 's/.*"frac_fp32_mfma_peak": \([0-9.]*\).*/\1/' | tr '\n' ' '; }
{
for lib in default ilp3 default ilp3; do
  if [ $lib = default ]; then unset CDX_LIB; else export CDX_LIB=$PWD/build_variants/libcdx_$lib.so; fi
  echo "== lib=$lib: config 4 / ChiTransformer / config 5: $(timeout 600 python tools/bench_configs.py cfg4:512 cfgT:1024:10 cfg5:16384 2>&1 | frac)"
  timeout 600 python tools/update_bench.py cfg2 cfg3 cfg4 cfg5 chitf 2>&1 | grep "update()" | cut -c1-110
done
} > $O/ilp_other_tus.txt 2>&1
cat $O/ilp_other_tus.txt
