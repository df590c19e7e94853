#!/bin/bash
# round 6: does the tile loop wrapper of the GEMM kernel (one iteration by default) change the default build?  prev = cdx_gemm.hip of commit 4dab819
O=gpurun_out/r6p; mkdir -p $O
{
for lib in default prev default prev; do
  if [ $lib = default ]; then unset CDX_LIB; else export CDX_LIB=$PWD/build_variants/libcdx_$lib.so; fi
  echo "== lib=$lib"
  timeout 300 python tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids | grep -v "adaLN\|final\|x_proj"
  timeout 600 python tools/bench_configs.py cfg4:512 cfg5:16384 cfgT:1024:10 2>&1 | grep -v "amdgpu.ids\|Warn" | cut -c1-60,150-260
done
} > $O/gemm_wrapper_ab.txt 2>&1
cat $O/gemm_wrapper_ab.txt
