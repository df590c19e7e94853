#!/bin/bash
# One libcdx build with extra -D flags for cdx_unet2.hip: tools/build_variant.sh <name> "<flags>" -> build_variants/libcdx_<name>.so
# (git-ignored, travels to the GPU box; selected with CDX_LIB=...).  The other objects come from the in-tree build.
# The in-tree build's own option for this file (__graft_entry__.EXTRA_FLAGS) is applied first; CDX_UNET2_BASE_FLAGS= (empty) drops it.
set -e
cd "$(dirname "$0")/.."
mkdir -p build_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I include ${CDX_UNET2_BASE_FLAGS--mllvm -amdgpu-sched-strategy=max-ilp} $2 -c cleandiffuser_amd/csrc/cdx_unet2.hip -o build_variants/cdx_unet2_$1.o
objs=$(ls cleandiffuser_amd/csrc/_obj/*.o | grep -v cdx_unet2.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs build_variants/cdx_unet2_$1.o -o build_variants/libcdx_$1.so
rm build_variants/cdx_unet2_$1.o
echo built build_variants/libcdx_$1.so
