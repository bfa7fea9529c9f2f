#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -k "dense or eigen or ladder or tridiag or two_stage or zero_modes or diagonal" > gpurun_out/r4_k9_pytest.log 2>&1; rc=$?; tail -2 gpurun_out/r4_k9_pytest.log; [ $rc -ne 0 ] && exit $rc
echo "== new"; python scratch/r4_k9_time.py 2>&1 | grep "^L="
echo "== old"; BODGE_AMD_LIBRARY=$GRAFT_REPO_ROOT/scratch/ab/libk9old.so python scratch/r4_k9_time.py 2>&1 | grep "^L="
echo "== new"; python scratch/r4_k9_time.py 2>&1 | grep "^L="
