#!/bin/bash
mkdir -p gpurun_out/suite
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/suite/gputests.log
cat gpurun_out/suite/gputests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
