#!/bin/bash
# round 6: -mllvm -amdgpu-sched-strategy=max-ilp for cdx_unet2.hip: the guided / large-batch / other-net rows, same box
O=gpurun_out/r6s; mkdir -p $O
{
for rep in 1 2; do
for lib in default ilp; do
  if [ $lib = default ]; then unset CDX_LIB; else export CDX_LIB=$PWD/build_variants/libcdx_$lib.so; fi
  echo "== lib=$lib"
  timeout 600 python tools/bench_configs.py cfg2g:256 cfg2g:3200 cfgKg:256 cfgAg:256 cfgAu:3200 cfg1 2>&1 | grep -v "amdgpu.ids\|Warn" | cut -c1-70,100-230
  timeout 300 python tools/time_cfg2.py 256 768 1536 2>&1 | grep -v amdgpu.ids | cut -c1-140
done
done
} > $O/sched_ilp_rows.txt 2>&1
cat $O/sched_ilp_rows.txt
export CDX_LIB=$PWD/build_variants/libcdx_ilp.so
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -2
