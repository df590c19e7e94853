#!/bin/bash
# round 5, call K: the 8-wave shape of the 128 x 128 GEMM tile (block sums over two K tiles, four waves per SIMD) against the 4-wave shape and
# against the sequential chain (kb0 build), same box; the GEMM-executor tests on the 8-wave shape.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5k
for v in "kb0 0" "cur 0" "cur 1" "kb0 0" "cur 0" "cur 1"; do
  set -- $v
  if [ "$1" = "kb0" ]; then export CDX_LIB=$GRAFT_REPO_ROOT/build_variants/libcdx_kb0.so; else unset CDX_LIB; fi
  export CDX_GEMM_W8=$2
  for cfg in cfg4:512 cfg3 cfgT:1024:10 cfg5:16384; do
    echo -n "$1 w8=$2 $cfg: "
    timeout 300 python tools/bench_configs.py $cfg 2>&1 | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(round(d['ms_per_call'], 2), 'ms', round(d.get('frac_fp32_mfma_peak', 0), 4))"
  done
done 2>&1 | tee gpurun_out/r5k/gemm_w8_ab.txt
unset CDX_LIB
export CDX_GEMM_W8=1
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short -k "gemm or dit or chiunet or chitf or resmlp or idql or pearcetf or cfg3 or cfg4 or cfg5 or heads or encoder or blocks or training or update or wgrad or linear or transformer or baseline" 2>&1 | tail -6 | tee gpurun_out/r5k/gpu_subset_w8.txt
timeout 600 python tools/dit_error_budget.py 2>&1 | grep -v amdgpu.ids | head -18 > gpurun_out/r5k/dit_error_budget_w8.txt; sed -n 3,8p gpurun_out/r5k/dit_error_budget_w8.txt
