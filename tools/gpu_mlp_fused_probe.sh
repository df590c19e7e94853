#!/bin/bash
# fused fc1 -> GELU -> fc2 probe (open item 1): correctness against a float64 host reference + time per call, four builds:
#   hipcc --offload-arch=gfx950 -O3 -DPROBE_BK2={8,16} -DPROBE_UPFRONT={0,1} -o tools/_bin/mlp_fused_probe_k${BK2}_u${UPFRONT} tools/mlp_fused_probe.hip
# (W2 slab of 8 k = two workgroups per CU, 16 k = one workgroup per CU and half the barriers; LDS operands of a slab requested up front or one
#  k pair ahead).  Binaries are built in the container (tools/_bin/ travels with the snapshot).
mkdir -p gpurun_out/mlpf
for v in k8_u0 k8_u1 k16_u0 k16_u1; do
    echo "## $v" >> gpurun_out/mlpf/mlp_fused_probe.txt
    timeout 60 tools/_bin/mlp_fused_probe_$v >> gpurun_out/mlpf/mlp_fused_probe.txt 2>&1
done
cat gpurun_out/mlpf/mlp_fused_probe.txt
