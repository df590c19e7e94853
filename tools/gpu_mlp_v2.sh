#!/bin/bash
# batch-tiled MLPs on the program kernel: the MLP fixtures, then the config-1 A/B that produced profiles/r03_cfg1_v2_mlp.txt (run when the
# round-1 kernel still existed: CDX_UNET2_MLP=0 selected it; now that hook means the PyTorch executor), then the guided entries
mkdir -p gpurun_out/r3u
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "fused_sample_matches or tile_mlp or pearce_mlp_widths or baseline_cfg1 or training_step or empty_and_ragged" 2>&1 | tail -25 > gpurun_out/r3u/mlp_tests.log
cat gpurun_out/r3u/mlp_tests.log
out=gpurun_out/r3u/cfg1_v2.txt
: > $out
for v in 1 0; do
  echo "== cfg1 CDX_UNET2_MLP=$v" >> $out
  CDX_UNET2_MLP=$v timeout 300 python tools/bench_configs.py cfg1 2>&1 | grep -v amdgpu.ids | cut -c1-400 >> $out
done
timeout 300 python tools/bench_configs.py cfg2g:256 cfg2g:3200 2>&1 | grep -v amdgpu.ids | cut -c1-300 >> $out
cat $out
