#!/bin/bash
# round 6, weight-gradient batches on a side stream: same-box A/B of CDX_TRAIN_WGRAD_SIDE (0 = at the end of the pass on its own stream)
# + the training parity tests under the new default.  -> gpurun_out/r6f/
O=gpurun_out/r6f; mkdir -p $O
export UPDATE_BENCH_GRAPH_ONLY=1
for rep in 1 2; do
for side in 0 8 16 32; do
  echo "== CDX_TRAIN_WGRAD_SIDE=$side (rep $rep)"
  CDX_TRAIN_WGRAD_SIDE=$side timeout 600 python tools/update_bench.py cfg2 cfg3 cfg4 cfg5 chitf 2>&1 | grep "update()"
done
done > $O/wgrad_side_ab.txt 2>&1
cat $O/wgrad_side_ab.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -k "update or train or graph or adam or classifier or wgrad or critic" 2>&1 | tail -8 > $O/tests.txt
cat $O/tests.txt
