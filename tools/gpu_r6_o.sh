#!/bin/bash
# round 6: persistent walk over tiles in the 8-wave GEMM kernel (-DCDX_GEMM_PERSIST=1: grid = resident slots, next tile's first K tile requested
# before the epilogue): per-shape micro-benchmark + configs 3 / 4 / 5 / ChiTransformer, same box; correctness: the GEMM / DiT / MLP parity tests
O=gpurun_out/r6o; mkdir -p $O
{
for lib in default persist default persist; do
  if [ $lib = default ]; then unset CDX_LIB; else export CDX_LIB=$PWD/build_variants/libcdx_$lib.so; fi
  echo "== lib=$lib"
  timeout 300 python tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids
done
for lib in default persist; do
  if [ $lib = default ]; then unset CDX_LIB; else export CDX_LIB=$PWD/build_variants/libcdx_$lib.so; fi
  echo "== lib=$lib"
  timeout 600 python tools/bench_configs.py cfg4:512 cfgT:1024:10 cfg5:16384 2>&1 | grep -v "amdgpu.ids\|Warn" | cut -c1-260
done
for slots in 256 768 1024; do
  echo "== lib=persist CDX_GEMM_PERSIST_SLOTS=$slots"
  CDX_GEMM_PERSIST_SLOTS=$slots timeout 300 python tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids | head -4
done
} > $O/gemm_persist_ab.txt 2>&1
cat $O/gemm_persist_ab.txt
export CDX_LIB=$PWD/build_variants/libcdx_persist.so
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "gemm or dit or mlp or linear or idql or chitf or transformer" 2>&1 | tail -3
