#!/bin/bash
mkdir -p gpurun_out/r4m
cd "$(dirname "$0")/.."
for v in single dual single dual; do
  echo "## $v"
  CDX_LIB=$PWD/build_variants/libcdx_$v.so timeout 200 python tools/time_cfg2.py 256 32 2>&1 | grep "^B="
done | tee gpurun_out/r4m/variants.txt
