#!/bin/bash
# round 6: group-in-registers GroupNorm backward + dy_possum (ABI 17): tests, then same-box A/B (CDX_GN_VEC=0 also switches the forward vec kernel off,
# CDX_TRAIN_POSSUM=0 the position sums)
O=gpurun_out/r6i; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "groupnorm" 2>&1 | tail -6 > $O/tests_gn.txt; cat $O/tests_gn.txt
export UPDATE_BENCH_GRAPH_ONLY=1
{
for rep in 1 2; do
  echo "== default (rep $rep)";                 timeout 300 python tools/update_bench.py cfg2 cfg3 2>&1 | grep "update()"
  echo "== CDX_TRAIN_POSSUM=0 (rep $rep)";      CDX_TRAIN_POSSUM=0 timeout 300 python tools/update_bench.py cfg2 2>&1 | grep "update()"
done
timeout 200 python tools/update_census.py cfg2 2>&1 | grep -v "Warning\|amdgpu.ids" | head -12
} > $O/gn_bwd_vec_ab.txt 2>&1
cat $O/gn_bwd_vec_ab.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -k "update or train or graph or adam or classifier or wgrad or critic or guided or gradient" 2>&1 | tail -6 > $O/tests.txt
cat $O/tests.txt
