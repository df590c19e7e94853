#!/bin/bash
# round 6: LLVM scheduling strategies for the two hot kernels (-mllvm -amdgpu-sched-strategy=...), same box
O=gpurun_out/r6r; mkdir -p $O
{
for rep in 1 2; do
for lib in default ilp memclause iterilp; do
  if [ $lib = default ]; then unset CDX_LIB; else export CDX_LIB=$PWD/build_variants/libcdx_$lib.so; fi
  echo "== program kernel lib=$lib"
  timeout 300 python tools/time_cfg2.py 256 3200 32 2>&1 | grep -v amdgpu.ids | cut -c1-140
done
done
for lib in default gilp gmemclause default gilp gmemclause; do
  if [ $lib = default ]; then unset CDX_LIB; else export CDX_LIB=$PWD/build_variants/libcdx_$lib.so; fi
  echo "== gemm lib=$lib"
  timeout 300 python tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids | grep -v "adaLN\|final\|x_proj"
  timeout 300 python tools/gemm_bench.py 4096,256,1280 2048,512,2560 2>&1 | grep -v amdgpu.ids
done
} > $O/sched_strategy_ab.txt 2>&1
cat $O/sched_strategy_ab.txt
