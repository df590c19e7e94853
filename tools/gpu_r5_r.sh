#!/bin/bash
# round 5, call R: rows per split-K slice of the weight-gradient kernel (CDX_WGRAD_MIN_CHUNKS x 16 rows): update() of configs 2 / 3 / 4 / transformer
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5r
export UPDATE_BENCH_GRAPH_ONLY=1
for v in 4 8 16 32 4 16; do
  export CDX_WGRAD_MIN_CHUNKS=$v
  echo "min_chunks=$v"; timeout 200 python tools/update_bench.py cfg2 cfg3 cfg4 chitf 2>&1 | grep "update()" | sed 's/^/   /'
done 2>&1 | tee gpurun_out/r5r/wgrad_min_chunks.txt
