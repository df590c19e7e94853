#!/bin/bash
# A/B of compile-time variants of the member kernel (build_variants/libcdx_<v>.so; tools/build_variant.sh), config 2 at B = 256 / 32.
#   VARIANTS="base merge" bash tools/gpu_r4_variants.sh
cd $GRAFT_REPO_ROOT
for round in 1 2; do
for v in $VARIANTS; do
  echo -n "$v: "
  CDX_LIB=$GRAFT_REPO_ROOT/build_variants/libcdx_$v.so timeout 300 python tools/time_cfg2.py 256 32 2>&1 | grep -v amdgpu.ids | cut -c1-150 | tr '\n' '|'
  echo
done
done 2>&1 | tee gpurun_out/r4variants_$TAG.txt
