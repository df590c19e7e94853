#!/bin/bash
O=gpurun_out/r6u; mkdir -p $O
{
for rep in 1 2; do
for lib in default mfmav1 nolive; do
  if [ $lib = default ]; then unset CDX_LIB; else export CDX_LIB=$PWD/build_variants/libcdx_$lib.so; fi
  echo "== program kernel lib=$lib"
  timeout 300 python tools/time_cfg2.py 256 768 32 2>&1 | grep -v amdgpu.ids | cut -c1-140
done
done
for lib in default gmfmav1 default gmfmav1; do
  if [ $lib = default ]; then unset CDX_LIB; else export CDX_LIB=$PWD/build_variants/libcdx_$lib.so; fi
  echo "== gemm lib=$lib"
  timeout 300 python tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids | grep -v "adaLN\|final\|x_proj\|big"
  timeout 300 python tools/gemm_bench.py 4096,256,1280 2048,512,2560 2>&1 | grep -v amdgpu.ids
done
} > $O/mfma_form_ab.txt 2>&1
cat $O/mfma_form_ab.txt
