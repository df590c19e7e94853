#!/bin/bash
# round 6: update() launch diet -- FiLM Linears as one node (_LinearMany), a ResidualBlock's two input paths as one node (_ConvPair): same-box A/B
O=gpurun_out/r6h; mkdir -p $O
export UPDATE_BENCH_GRAPH_ONLY=1
{
for rep in 1 2; do
  echo "== default (rep $rep)";                      timeout 300 python tools/update_bench.py cfg2 2>&1 | grep "update()"
  echo "== CDX_TRAIN_FILM_BATCH=0 (rep $rep)";       CDX_TRAIN_FILM_BATCH=0 timeout 300 python tools/update_bench.py cfg2 2>&1 | grep "update()"
  echo "== CDX_TRAIN_CONV_PAIR=0 (rep $rep)";        CDX_TRAIN_CONV_PAIR=0 timeout 300 python tools/update_bench.py cfg2 2>&1 | grep "update()"
  echo "== both off (rep $rep)";                     CDX_TRAIN_FILM_BATCH=0 CDX_TRAIN_CONV_PAIR=0 timeout 300 python tools/update_bench.py cfg2 2>&1 | grep "update()"
done
timeout 200 python tools/update_census.py cfg2 2>&1 | grep -v "Warning\|amdgpu.ids" | head -45
} > $O/launch_diet_ab.txt 2>&1
cat $O/launch_diet_ab.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -k "update or train or graph or adam or classifier or wgrad or critic" 2>&1 | tail -8 > $O/tests.txt
cat $O/tests.txt
