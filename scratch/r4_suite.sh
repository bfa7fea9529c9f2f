#!/bin/bash
# the whole GPU suite as the driver runs it, with the slowest tests listed
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( while true; do sleep 60; echo "[suite] $(date +%T) $(tail -c 200 gpurun_out/r04_pytest_gpu.log | tr '\n' ' ' | tail -c 120)"; done ) &
TICK=$!
timeout -k 10 1100 python -m pytest tests/ -x -q -m gpu --durations=40 > gpurun_out/r04_pytest_gpu.log 2>&1
RC=$?
kill $TICK
tail -60 gpurun_out/r04_pytest_gpu.log
exit $RC
