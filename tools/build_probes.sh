#!/bin/bash
# Build the stand-alone gfx950 probes into tools/_bin/ (git-ignored; the directory travels to the GPU box with the gpurun snapshot).
# Run in the build container before a `gpurun -- bash tools/gpu_*_probe.sh` call.
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_bin
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
for p in l2_stream_probe xwg_exchange_probe lds_dma_probe; do
    $HIPCC --offload-arch=gfx950 -O3 -o tools/_bin/$p tools/$p.hip
done
for k in 8 16; do
    for u in 0 1; do
        $HIPCC --offload-arch=gfx950 -O3 -DPROBE_BK2=$k -DPROBE_UPFRONT=$u -o tools/_bin/mlp_fused_probe_k${k}_u${u} tools/mlp_fused_probe.hip
    done
done
cp tools/_bin/mlp_fused_probe_k8_u0 tools/_bin/mlp_fused_probe
ls -la tools/_bin
