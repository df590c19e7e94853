#!/bin/bash
# round 6: finer split-K for the small conv GEMMs of a training step (slices of >= 4 K tiles instead of >= 8; scratch from K >= 128 instead of 256)
O=gpurun_out/r6k; mkdir -p $O
export UPDATE_BENCH_GRAPH_ONLY=1
{
for cfg in "8 256" "4 256" "4 128" "8 128" "2 64" "8 256" "4 128"; do
  set -- $cfg
  echo "== CDX_GEMM_SPLITK_MIN_TILES=$1 CDX_TRAIN_SPLITK_MINK=$2"
  CDX_GEMM_SPLITK_MIN_TILES=$1 CDX_TRAIN_SPLITK_MINK=$2 timeout 300 python tools/update_bench.py cfg2 cfg3 cfg5 2>&1 | grep "update()"
  CDX_GEMM_SPLITK_MIN_TILES=$1 CDX_TRAIN_SPLITK_MINK=$2 timeout 200 python tools/update_census.py cfg2 2>&1 | grep "cdx_gemm_kernel<true, 1, true\|splitk\|launches,"
done
} > $O/splitk_fine.txt 2>&1
cat $O/splitk_fine.txt
