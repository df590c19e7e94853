#!/bin/bash
# round 6: library variants x CDX_UNET2_DEFER (one process per setting: the compiled program is cached per process)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6e
for v in $VARIANTS; do
  for d in ${DEFERS:-0 1}; do
    [ $v = d0 ] && [ $d = 1 ] && continue
    echo "$v defer=$d $(CDX_LIB=$GRAFT_REPO_ROOT/build_variants/libcdx_$v.so CDX_UNET2_DEFER=$d timeout 300 python tools/time_cfg2.py 256 2>&1 | grep traj/s | cut -c50-130)"
  done
done | tee gpurun_out/r6e/ab_${TAG:-x}.txt
if [ -n "$CHECKV" ]; then CDX_LIB=$GRAFT_REPO_ROOT/build_variants/libcdx_$CHECKV.so timeout 200 python tools/dbg_group.py 256 2>&1 | grep -v amdgpu.ids | tail -3; fi
if [ -n "$PROFILE" ]; then CDX_LIB=$GRAFT_REPO_ROOT/build_variants/libcdx_$PROFILE.so timeout 200 python tools/op_profile2.py 256 group4 2>&1 | grep -v amdgpu.ids > gpurun_out/r6e/op_profile_group4_$PROFILE.txt; tail -1 gpurun_out/r6e/op_profile_group4_$PROFILE.txt; fi
