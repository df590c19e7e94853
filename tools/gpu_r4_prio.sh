#!/bin/bash
cd $GRAFT_REPO_ROOT
for f in 0 2 4; do
  for cfg in cfg4:512 cfg3 cfgT:1024:10 cfg5:16384; do
    echo -n "XCD_ORDER(prio bits)=$f $cfg: "
    CDX_GEMM_XCD_ORDER=$f timeout 300 python tools/bench_configs.py $cfg 2>&1 | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(round(d['ms_per_call'], 2), 'ms', round(d.get('frac_fp32_mfma_peak', 0), 4))"
  done
done 2>&1 | tee gpurun_out/r4prio.txt
