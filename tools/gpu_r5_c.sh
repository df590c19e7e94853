#!/bin/bash
# round 5, call C: the final K-blocked GEMM (simple two-chain form, packed flush) -- same-box A/B, the DiT error budget, the whole GPU
# suite with the tightened config-4 / fp64 tests, the ChiUNet1d training nodes, the repair launch + fault hook, smoke.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5c
for v in kb0 kb1 kb0 kb1; do
  export CDX_LIB=$GRAFT_REPO_ROOT/build_variants/libcdx_$v.so
  for cfg in cfg4:512 cfg3 cfgT:1024:10 cfg5:16384; do
    echo -n "$v $cfg: "
    timeout 300 python tools/bench_configs.py $cfg 2>&1 | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(round(d['ms_per_call'], 2), 'ms', round(d.get('frac_fp32_mfma_peak', 0), 4))"
  done
done 2>&1 | tee gpurun_out/r5c/gemm_kblock_ab.txt
unset CDX_LIB
timeout 600 python tools/dit_error_budget.py > gpurun_out/r5c/dit_error_budget.txt 2>&1
tail -8 gpurun_out/r5c/dit_error_budget.txt
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short -s 2>&1 > gpurun_out/r5c/gpu_suite_full.txt
grep -E "native vs fp64|beyond the elementwise|^(FAILED|ERROR)|passed|failed" gpurun_out/r5c/gpu_suite_full.txt | head -60
grep -B2 -A25 "^___" gpurun_out/r5c/gpu_suite_full.txt | head -150 > gpurun_out/r5c/gpu_suite_failures.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5c/smoke.txt
