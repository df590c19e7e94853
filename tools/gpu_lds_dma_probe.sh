#!/bin/bash
# direct-to-LDS prefetch of the next op's record head next to an epilogue-like load (design input for the LDS-fed stream head)
mkdir -p gpurun_out/ldsdma
timeout 100 tools/_bin/lds_dma_probe > gpurun_out/ldsdma/lds_dma_probe.txt 2>&1
cat gpurun_out/ldsdma/lds_dma_probe.txt
