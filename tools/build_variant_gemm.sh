#!/bin/bash
# tools/build_variant_gemm.sh <name> "<flags>": libcdx with cdx_gemm.hip compiled with extra -D flags -> build_variants/libcdx_<name>.so
set -e
cd "$(dirname "$0")/.."
mkdir -p build_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I include $2 -c cleandiffuser_amd/csrc/cdx_gemm.hip -o build_variants/cdx_gemm_$1.o
objs=$(ls cleandiffuser_amd/csrc/_obj/*.o | grep -v cdx_gemm.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs build_variants/cdx_gemm_$1.o -o build_variants/libcdx_$1.so
rm build_variants/cdx_gemm_$1.o
echo built build_variants/libcdx_$1.so
