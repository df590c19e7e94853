#!/bin/bash
# fused fc1 -> GELU -> fc2 probe (open item 1): correctness against a float64 host reference + time per call
mkdir -p gpurun_out/mlpf
timeout 60 tools/_bin/mlp_fused_probe > gpurun_out/mlpf/mlp_fused_probe.txt 2>&1
cat gpurun_out/mlpf/mlp_fused_probe.txt
