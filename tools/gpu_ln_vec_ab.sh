#!/bin/bash
# vectorised LayerNorm: parity of everything that uses it, then config 4 / ChiTransformer A/B
mkdir -p gpurun_out/r3y
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "bigbatch or dit or chitf or transformer or idql or resmlp or baseline_cfg4 or baseline_cfg5 or pearcetf or heads or condition_encoders" 2>&1 | tail -8 > gpurun_out/r3y/tests.log
cat gpurun_out/r3y/tests.log
out=gpurun_out/r3y/ln_vec_ab.txt
: > $out
for v in 0 1 0 1; do
  echo "== CDX_LN_VEC=$v" >> $out
  CDX_LN_VEC=$v timeout 300 python tools/bench_configs.py cfg4 cfgT 2>&1 | grep -v amdgpu.ids | cut -c1-260 >> $out
done
cat $out
