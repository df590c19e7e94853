#!/bin/bash
mkdir -p gpurun_out/r3s
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/r3s/gputests.log
cat gpurun_out/r3s/gputests.log
