#!/bin/bash
# round 6: the deferred exchange (program: CDX_UNET2_DEFER, kernel: build_variants/libcdx_d1.so) against the committed state (libcdx_d0.so):
# correctness against the ordinary program first, then timings in separate processes (the compiled program is cached per process)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6d2
export CDX_LIB=$GRAFT_REPO_ROOT/build_variants/libcdx_d1.so
timeout 200 python tools/dbg_group.py 256 2>&1 | grep -v amdgpu.ids | tail -3
for rep in 1 2; do
  CDX_LIB=$GRAFT_REPO_ROOT/build_variants/libcdx_d0.so CDX_UNET2_DEFER=0 timeout 300 python tools/time_cfg2.py 256 2>&1 | grep traj/s | cut -c1-140 | sed 's/^/d0 nodefer  /'
  CDX_UNET2_DEFER=0 timeout 300 python tools/time_cfg2.py 256 2>&1 | grep traj/s | cut -c1-140 | sed 's/^/d1 nodefer  /'
  timeout 300 python tools/time_cfg2.py 256 2>&1 | grep traj/s | cut -c1-140 | sed 's/^/d1 defer    /'
done 2>&1 | tee gpurun_out/r6d2/ab_defer.txt
if [ -n "$PROFILE" ]; then timeout 200 python tools/op_profile2.py 256 group4 2>&1 | grep -v amdgpu.ids > gpurun_out/r6d2/op_profile_group4_defer.txt; tail -1 gpurun_out/r6d2/op_profile_group4_defer.txt; fi
if [ -n "$CHECK" ]; then timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "unet2 or three_traj or test_fused_sample_matches_reference_fixture or group or split" 2>&1 | tail -3; fi
