#!/bin/bash
# round 6: compile-time LDS stage in the GEMM K loops (8-wave and 64 x 64 shapes) against the previous commit's source (lib=head), same box
O=gpurun_out/r6q; mkdir -p $O
export UPDATE_BENCH_GRAPH_ONLY=1
{
for lib in default head default head; do
  if [ $lib = default ]; then unset CDX_LIB; else export CDX_LIB=$PWD/build_variants/libcdx_$lib.so; fi
  echo "== lib=$lib"
  timeout 300 python tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids | grep -v "adaLN\|final\|x_proj"
  timeout 300 python tools/gemm_bench.py 4096,256,1280 8192,64,320 2048,512,2560 1024,1024,5120 2>&1 | grep -v amdgpu.ids
  timeout 600 python tools/bench_configs.py cfg3 cfg4:512 cfg5:16384 cfgT:1024:10 2>&1 | grep -v "amdgpu.ids\|Warn" | cut -c1-60,150-260
  timeout 300 python tools/update_bench.py cfg2 cfg3 cfg5 2>&1 | grep "update()"
done
} > $O/gemm_static_stage_ab.txt 2>&1
cat $O/gemm_static_stage_ab.txt
unset CDX_LIB
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "gemm or dit or mlp or linear or idql or chitf or transformer or chiunet or conv or update" 2>&1 | tail -3
