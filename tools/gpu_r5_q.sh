#!/bin/bash
# round 5, call Q: float4 LayerNorm kernel (CDX_LN_VEC) and the split-N remainder GEMM on a side stream (CDX_GEMM_SPLIT_N_SIDE), each on / off
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5q
run() { timeout 300 python tools/bench_configs.py $1 2>&1 | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(round(d['ms_per_call'], 2), 'ms', round(d.get('frac_fp32_mfma_peak', 0), 4))" 2>&1 | tail -1; }
for v in "0 0" "1 0" "0 1" "1 1" "0 0" "1 1"; do
  set -- $v
  export CDX_LN_VEC=$1 CDX_GEMM_SPLIT_N_SIDE=$2   # (the side-stream path was removed after this run)
  echo -n "ln_vec=$1 side=$2 cfg4:512: "; run cfg4:512
done 2>&1 | tee gpurun_out/r5q/ab.txt
export CDX_GEMM_SPLIT_N_SIDE=1
for v in 0 1 0 1; do
  export CDX_LN_VEC=$v
  for cfg in cfgT:1024:10 cfg5:16384; do echo -n "ln_vec=$v $cfg: "; run $cfg; done
done 2>&1 | tee -a gpurun_out/r5q/ab.txt
unset CDX_LN_VEC CDX_GEMM_SPLIT_N_SIDE
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dit or chitf or idql or cfg4 or cfg5 or layernorm or transformer or resmlp or gemm" 2>&1 | tail -4 | tee gpurun_out/r5q/tests.txt
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "update or relayout" 2>&1 | tail -3 | tee -a gpurun_out/r5q/tests.txt
timeout 300 python tools/update_bench.py cfg2 2>&1 | grep "update()" | tee gpurun_out/r5q/update_cfg2.txt
