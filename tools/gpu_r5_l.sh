#!/bin/bash
# round 5, call L: final same-box A/B of the GEMM kernel -- sequential chain (kb0 build) / K-blocked 4-wave (CDX_GEMM_W8=0) / K-blocked 8-wave
# (the default) --, the DiT error budget and the whole GPU suite on the default.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5l
for v in "kb0 0" "cur 0" "cur 1" "kb0 0" "cur 0" "cur 1"; do
  set -- $v
  if [ "$1" = "kb0" ]; then export CDX_LIB=$GRAFT_REPO_ROOT/build_variants/libcdx_kb0.so; else unset CDX_LIB; fi
  export CDX_GEMM_W8=$2
  for cfg in cfg4:512 cfg3 cfgT:1024:10 cfg5:16384; do
    echo -n "$1 w8=$2 $cfg: "
    timeout 300 python tools/bench_configs.py $cfg 2>&1 | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(round(d['ms_per_call'], 2), 'ms', round(d.get('frac_fp32_mfma_peak', 0), 4))" 2>&1 | tail -1
  done
done 2>&1 | tee gpurun_out/r5l/gemm_w8_ab.txt
unset CDX_LIB CDX_GEMM_W8
timeout 600 python tools/dit_error_budget.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r5l/dit_error_budget.txt; tail -6 gpurun_out/r5l/dit_error_budget.txt
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short -s 2>&1 > gpurun_out/r5l/gpu_suite_full.txt
grep -E "native vs fp64|beyond the elementwise|^(FAILED|ERROR)|passed|failed" gpurun_out/r5l/gpu_suite_full.txt | cut -c1-400 | head -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "amdgpu.ids\|sync_debug\|Synchronization" | tee gpurun_out/r5l/smoke.txt
