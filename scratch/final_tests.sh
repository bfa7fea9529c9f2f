#!/bin/bash
# round-end style validation: the whole GPU suite, then smoke(); log under gpurun_out/final/
set -o pipefail
OUT=$PWD/gpurun_out/final; mkdir -p $OUT
timeout -k 10 1050 python3 -m pytest tests -m gpu -x -q --durations=8 2>&1 | tee $OUT/pytest_gpu.log
timeout -k 10 120 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tee $OUT/smoke.log
