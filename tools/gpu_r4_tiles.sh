#!/bin/bash
# Round 4: 64 x 64 tiles without split-K against 128 x 128 tiles + split-K on the 256-tile conv layers of config 3 (and what the
# threshold does to configs 4 / 5 / ChiTransformer)
cd $GRAFT_REPO_ROOT
for t in 192 300 520 1100; do
  for cfg in cfg3 cfg3:256 cfg4:512 cfgT:1024:10; do
    echo -n "SMALL_TILE_BELOW=$t $cfg: "
    CDX_GEMM_SMALL_TILE_BELOW=$t timeout 300 python tools/bench_configs.py $cfg 2>&1 | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(round(d['ms_per_call'], 2), 'ms', round(d.get('frac_fp32_mfma_peak', 0), 4))"
  done
done 2>&1 | tee gpurun_out/r4tiles.txt
