# round-end check (GPU box): full GPU suite, smoke, the driver-style bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fin
( time timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 ) 2>&1 | tail -9
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
( time timeout 900 python bench.py > gpurun_out/fin/r02_bench_n1.json 2> gpurun_out/fin/bench.err ) 2>&1 | tail -4
python - <<'PY'
import json
d = json.loads(open('gpurun_out/fin/r02_bench_n1.json').read().strip().splitlines()[-1])
print(round(d['value']), d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])
for o in d['other_configs']:
    print(o['name'], o.get('error') or (round(o['value']), round(o['ms_per_call'], 2), round(o['roofline_frac'], 4)))
PY
