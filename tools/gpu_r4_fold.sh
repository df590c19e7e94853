#!/bin/bash
# Round 4: GroupNorm-folded split-K reduction (config 3 A/B)
cd $GRAFT_REPO_ROOT
E=gpurun_out/r4fold
mkdir -p $E
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "chiunet or cfg3 or gemm or unet_attention or local_cond or janner_big or full_size" 2>&1 | tail -5 > $E/tests.log; tail -3 $E/tests.log
for f in 1 0; do
  CDX_UNET_GN_FOLD=$f timeout 300 python tools/bench_configs.py cfg3 2>&1 | tail -1 > $E/cfg3_fold$f.json; cut -c1-400 $E/cfg3_fold$f.json
done
CDX_UNET_GN_FOLD=1 timeout 300 python tools/bench_configs.py cfg3:256 2>&1 | tail -1 | cut -c1-300
CDX_UNET_GN_FOLD=0 timeout 300 python tools/bench_configs.py cfg3:256 2>&1 | tail -1 | cut -c1-300
