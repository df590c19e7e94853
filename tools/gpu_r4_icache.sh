#!/bin/bash
# Instruction-cache counters of the program kernel's instantiations: headline (grouped member kernel, 49 KB of code), guided (BWD, 88 KB),
# config 1 (MLP, 121 KB).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
E=$R/gpurun_out/r4icache
mkdir -p $E
rocprofv3 --list-avail 2>/dev/null | grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQ_WAIT_INST[A-Z_]*" | sort -u | tr '\n' ' '; echo
for cfg in cfg2g:256 cfg1 ; do
  tag=$(echo $cfg | tr ':' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES --output-format csv -d $E/$tag -- python $R/tools/bench_configs.py $cfg > $E/$tag.log 2>&1
  f=$(find $E/$tag -name "*counter_collection.csv" | head -1)
  python - "$f" $cfg <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
try:
    for row in csv.DictReader(open(sys.argv[1])):
        if 'cdx_unet2_kernel' in row.get('Kernel_Name', ''):
            acc[(row['Kernel_Name'][30:95], row['Counter_Name'])].append(float(row['Counter_Value']))
except Exception as e:
    print("pmc parse failed", e)
for k, v in sorted(acc.items()):
    print(sys.argv[2], "PMC", k[1], "mean_per_dispatch", sum(v) / len(v), "dispatches", len(v), "kernel", k[0])
PY
  rm -rf $E/$tag
done
timeout 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES --output-format csv -d $E/head -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs > $E/head.log 2>&1
f=$(find $E/head -name "*counter_collection.csv" | head -1)
python - "$f" headline <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    if 'cdx_unet2_kernel' in row.get('Kernel_Name', ''):
        acc[(row['Kernel_Name'][30:95], row['Counter_Name'])].append(float(row['Counter_Value']))
for k, v in sorted(acc.items()):
    print(sys.argv[2], "PMC", k[1], "mean_per_dispatch", sum(v) / len(v), "dispatches", len(v), "kernel", k[0])
PY
rm -rf $E/head
