#!/bin/bash
cd $GRAFT_REPO_ROOT
for f in 50 51 76 101; do
  for cfg in cfg3 cfg3:256 cfg3:64; do
    echo -n "SPLITK_FILL=$f $cfg: "
    CDX_GEMM_SPLITK_FILL=$f timeout 300 python tools/bench_configs.py $cfg 2>&1 | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(round(d['ms_per_call'], 2), 'ms', round(d.get('frac_fp32_mfma_peak', 0), 4))"
  done
done 2>&1 | tee gpurun_out/r4fill.txt
