#!/bin/bash
# round 6: split-K slice count rounded down (never more workgroups than resident slots) vs up: config 3, training steps, guided per-step executor
O=gpurun_out/r6ad; mkdir -p $O
frac() { grep -v "amdgpu.ids\|Warn" | sed 's/.*"frac_fp32_mfma_peak": \([0-9.]*\).*/\1/' | tr '\n' ' '; }
export UPDATE_BENCH_GRAPH_ONLY=1
{
for rep in 1 2; do
for rd in ceil floor; do
  export CDX_GEMM_SPLITK_ROUND=$rd
  echo "round=$rd: config 3 $(timeout 300 python tools/bench_configs.py cfg3 2>&1 | frac) config 3 B=130 $(timeout 300 python tools/bench_configs.py cfg3:130 2>&1 | frac)"
  timeout 300 python tools/update_bench.py cfg2 cfg3 cfg4 cfg5 chitf 2>&1 | grep "update()" | sed "s/^/   /" | cut -c1-120
done
done
} > $O/splitk_round.txt 2>&1
cat $O/splitk_round.txt
