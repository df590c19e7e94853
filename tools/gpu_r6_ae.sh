#!/bin/bash
# round 6: pokes at the 64 x 64 GEMM shape (compiled occupancy, loop unrolling), same box
O=gpurun_out/r6ae; mkdir -p $O
export UPDATE_BENCH_GRAPH_ONLY=1
{
for lib in default occ3 occ2 occ5 unr1 default; do
  if [ $lib = default ]; then unset CDX_LIB; else export CDX_LIB=$PWD/build_variants/libcdx_$lib.so; fi
  echo "== lib=$lib"
  timeout 300 python tools/gemm_bench.py 4096,256,1280 2048,512,2560 8192,64,320 1024,1024,5120 16384,256,256 2>&1 | grep -v amdgpu.ids
  timeout 300 python tools/update_bench.py cfg2 2>&1 | grep "update()"
done
} > $O/small_tile_pokes.txt 2>&1
cat $O/small_tile_pokes.txt
