#!/bin/bash
# round 6: samples per wave of the register GroupNorm backward (atomics per address = B / (4 spw)): sweep on config 2 / config 3
O=gpurun_out/r6j; mkdir -p $O
export UPDATE_BENCH_GRAPH_ONLY=1
{
for spw in 1 2 4 8 1 4; do
  echo "== CDX_GN_BWD_SPW=$spw"; CDX_GN_BWD_SPW=$spw timeout 300 python tools/update_bench.py cfg2 cfg3 2>&1 | grep "update()"
  CDX_GN_BWD_SPW=$spw timeout 200 python tools/update_census.py cfg2 2>&1 | grep "groupnorm_bwd\|launches,"
done
} > $O/gn_bwd_spw.txt 2>&1
cat $O/gn_bwd_spw.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "groupnorm" 2>&1 | tail -3
CDX_GN_BWD_SPW=4 timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "groupnorm" 2>&1 | tail -3
