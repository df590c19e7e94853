#!/bin/bash
# round 5, call M: the GEMM A/B table once more with a kb0 build of the FINAL source (sequential chain / K-blocked 4 waves / K-blocked 8 waves).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5m
for v in "kb0 0" "cur 0" "cur 1" "kb0 0" "cur 0" "cur 1"; do
  set -- $v
  if [ "$1" = "kb0" ]; then export CDX_LIB=$GRAFT_REPO_ROOT/build_variants/libcdx_kb0.so; else unset CDX_LIB; fi
  export CDX_GEMM_W8=$2
  for cfg in cfg4:512 cfg3 cfgT:1024:10 cfg5:16384; do
    echo -n "$1 w8=$2 $cfg: "
    timeout 300 python tools/bench_configs.py $cfg 2>&1 | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(round(d['ms_per_call'], 2), 'ms', round(d.get('frac_fp32_mfma_peak', 0), 4))" 2>&1 | tail -1
  done
done 2>&1 | tee gpurun_out/r5m/gemm_w8_ab.txt
