#!/bin/bash
# GEMM kernel variants (build_variants/libcdx_<v>.so, tools/build_variant_gemm.sh): workgroup timeline of two DiT shapes, the GEMM-executor
# configs, and the GEMM / executor tests of the GPU suite on the CHECK variant.
cd $GRAFT_REPO_ROOT
for v in $VARIANTS; do
  export CDX_LIB=$GRAFT_REPO_ROOT/build_variants/libcdx_$v.so
  for shape in "65536 1280 320 gelu_tanh" "65536 256 320" "65536 1024 1024"; do
    echo -n "$v: "; timeout 120 python tools/gemm_trace.py $shape 2>&1 | grep -A3 "^M=" | tr '\n' ' ' | cut -c1-330; echo
  done
  for cfg in cfg4:512 cfg3 cfgT:1024:10 cfg5:16384 cfg3:256; do
    echo -n "$v $cfg: "
    timeout 300 python tools/bench_configs.py $cfg 2>&1 | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(round(d['ms_per_call'], 2), 'ms', round(d.get('frac_fp32_mfma_peak', 0), 4))"
  done
done 2>&1 | tee gpurun_out/r4gemm_ab_$TAG.txt
unset CDX_LIB
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short -x -k "gemm or dit or chiunet or chitf or resmlp or idql or pearcetf or cfg3 or cfg4 or cfg5 or heads or encoder or blocks or training or update or wgrad or linear" 2>&1 | tail -4
