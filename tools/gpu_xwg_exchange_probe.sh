#!/bin/bash
# cost of an in-launch cross-workgroup exchange (what a small-batch split of the program kernel would pay per layer)
mkdir -p gpurun_out/xwg
timeout 120 tools/_bin/xwg_exchange_probe > gpurun_out/xwg/xwg_exchange_probe.txt 2>&1
cat gpurun_out/xwg/xwg_exchange_probe.txt
