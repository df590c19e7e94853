#!/bin/bash
# ChiUNet1d on the program kernel: parity tests, then the small-batch A/B that produced profiles/r03_chiunet_v2_small_batch.txt (run when the
# round-1 kernel still existed: CDX_UNET2=0 selected it; since its deletion that hook sends the request to the GEMM executor)
mkdir -p gpurun_out/r3r
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "chiunet or baseline_cfg3 or fused_sample_matches or training_step" 2>&1 | tail -25 > gpurun_out/r3r/chi_tests.log
cat gpurun_out/r3r/chi_tests.log
out=gpurun_out/r3r/chiunet_v2_small_batch.txt
: > $out
for spec in cfg3:8:32 cfg3:64:32 cfg3:8:64 cfg3:64:64; do
  for v2 in 1 0; do
    echo "== $spec CDX_UNET2=$v2" >> $out
    CDX_UNET2=$v2 timeout 300 python tools/bench_configs.py $spec 2>&1 | grep -v amdgpu.ids | cut -c1-330 >> $out
  done
done
cat $out
