#!/bin/bash
mkdir -p gpurun_out/r4q
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r4q/pytest_full.log 2>&1
grep -n "Fatal\|Memory access\|Aborted\|FAILED" gpurun_out/r4q/pytest_full.log | head -20
tail -4 gpurun_out/r4q/pytest_full.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -6
