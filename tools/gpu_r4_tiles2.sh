#!/bin/bash
cd $GRAFT_REPO_ROOT
for cfg in cfgT:1024:10 cfg4:512 cfg3 cfg5:16384 cfg3:64; do
  echo -n "$cfg: "
  timeout 300 python tools/bench_configs.py $cfg 2>&1 | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(round(d['ms_per_call'], 2), 'ms', round(d.get('frac_fp32_mfma_peak', 0), 4))"
done
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short -x -k "gemm or dit or chiunet or chitf or resmlp or idql or pearcetf or cfg3 or cfg4 or cfg5 or heads or encoder or blocks or training or update or linear or guided" 2>&1 | tail -3
