#!/bin/bash
# round 5, call N: 8-wave GEMM shape, K tile 16 (two tiles per flush; default) against K tile 32 (one tile per flush; -DCDX_GEMM_W8_BK=32).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5n
CDX_LIB=$GRAFT_REPO_ROOT/build_variants/libcdx_bk32.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "gemm or linear or cfg4 or dit or chitf or conv" 2>&1 | tail -3 | tee gpurun_out/r5n/bk32_tests.txt
for v in bk16 bk32 bk16 bk32; do
  if [ "$v" = "bk32" ]; then export CDX_LIB=$GRAFT_REPO_ROOT/build_variants/libcdx_bk32.so; else unset CDX_LIB; fi
  for cfg in cfg4:512 cfg3 cfgT:1024:10 cfg5:16384; do
    echo -n "$v $cfg: "
    timeout 300 python tools/bench_configs.py $cfg 2>&1 | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(round(d['ms_per_call'], 2), 'ms', round(d.get('frac_fp32_mfma_peak', 0), 4))" 2>&1 | tail -1
  done
done 2>&1 | tee gpurun_out/r5n/gemm_bk_ab.txt
