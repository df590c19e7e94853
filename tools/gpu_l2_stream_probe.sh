#!/bin/bash
# L2 -> CU stream probe (no arithmetic): the ceiling of the program kernel's weight stream
mkdir -p gpurun_out/r3p
timeout 300 tools/_bin/l2_stream_probe > gpurun_out/r3p/l2_stream_probe.txt 2>&1
cat gpurun_out/r3p/l2_stream_probe.txt
