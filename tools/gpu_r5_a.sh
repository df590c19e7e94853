#!/bin/bash
# round 5, call A: K-blocked accumulation of cdx_gemm_kernel -- same-box A/B of the GEMM executors (kb0 = sequential chain, kb1 = blocked),
# the DiT error budget on the blocked kernel, then the whole GPU suite on the default library.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5a
for v in kb0 kb1 kb0 kb1; do
  export CDX_LIB=$GRAFT_REPO_ROOT/build_variants/libcdx_$v.so
  for cfg in cfg4:512 cfg3 cfgT:1024:10 cfg5:16384; do
    echo -n "$v $cfg: "
    timeout 300 python tools/bench_configs.py $cfg 2>&1 | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(round(d['ms_per_call'], 2), 'ms', round(d.get('frac_fp32_mfma_peak', 0), 4))"
  done
done 2>&1 | tee gpurun_out/r5a/gemm_kblock_ab.txt
unset CDX_LIB
timeout 600 python tools/dit_error_budget.py > gpurun_out/r5a/dit_error_budget.txt 2>&1
tail -25 gpurun_out/r5a/dit_error_budget.txt
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short -x 2>&1 | tail -15 | tee gpurun_out/r5a/gpu_suite.txt
