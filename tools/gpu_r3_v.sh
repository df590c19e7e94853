#!/bin/bash
# one program kernel left: full GPU suite + smoke
mkdir -p gpurun_out/r3v
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -60 > gpurun_out/r3v/gputests.log
cat gpurun_out/r3v/gputests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
