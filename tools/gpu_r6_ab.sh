#!/bin/bash
# round 6: same-box A/B of kernel variants (build_variants/libcdx_<v>.so, tools/build_variant.sh): config 2 at B = 256 / 32 / 512 / 3200
#   VARIANTS="base new" bash tools/gpu_r6_ab.sh       (CHECK=<v>: the program-kernel parity subset on that variant; PROFILE=<v>: op profile)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6ab
for rep in 1 2; do
for v in $VARIANTS; do
  echo "== $v (pass $rep)"
  CDX_LIB=$GRAFT_REPO_ROOT/build_variants/libcdx_$v.so timeout 300 python tools/time_cfg2.py ${BATCHES:-256 32 512 3200} 2>&1 | grep "traj/s" | cut -c1-150
done
done 2>&1 | tee gpurun_out/r6ab/ab_${TAG:-x}.txt
if [ -n "$CHECK" ]; then
  CDX_LIB=$GRAFT_REPO_ROOT/build_variants/libcdx_$CHECK.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "unet2 or three_traj or test_fused_sample_matches_reference_fixture or group or split or guided" 2>&1 | tail -3
fi
for v in $PROFILE; do
  CDX_LIB=$GRAFT_REPO_ROOT/build_variants/libcdx_$v.so timeout 200 python tools/op_profile2.py 256 group4 2>&1 | grep -v amdgpu.ids > gpurun_out/r6ab/op_profile_group4_$v.txt
  tail -1 gpurun_out/r6ab/op_profile_group4_$v.txt
done
