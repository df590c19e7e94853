#!/bin/bash
# round 5, call B: (1) pipelined K-blocked GEMM (kb2) against the sequential chain (kb0), same box; (2) what the repair launch costs;
# (3) the whole GPU suite on the default library (K-blocked GEMM, repair launch, device query), all failures listed.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5b
for v in kb0 kb2 kb0 kb2; do
  export CDX_LIB=$GRAFT_REPO_ROOT/build_variants/libcdx_$v.so
  for cfg in cfg4:512 cfg3 cfgT:1024:10 cfg5:16384; do
    echo -n "$v $cfg: "
    timeout 300 python tools/bench_configs.py $cfg 2>&1 | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(round(d['ms_per_call'], 2), 'ms', round(d.get('frac_fp32_mfma_peak', 0), 4))"
  done
done 2>&1 | tee gpurun_out/r5b/gemm_kblock_ab.txt
unset CDX_LIB
timeout 300 python tools/time_cfg2.py 256 256:CDX_UNET2_REPAIR=0 256 256:CDX_UNET2_REPAIR=0 32 32:CDX_UNET2_REPAIR=0 32:CDX_UNET2_SPLIT_SYNC=1 128 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5b/repair_cost.txt
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short 2>&1 | tail -150 > gpurun_out/r5b/gpu_suite.txt
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r5b/gpu_suite.txt | head -60
