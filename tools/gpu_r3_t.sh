#!/bin/bash
mkdir -p gpurun_out/r3t
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -60 > gpurun_out/r3t/gputests.log
cat gpurun_out/r3t/gputests.log
