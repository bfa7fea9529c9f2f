#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not dense and not eigen and not jacobi and not ladder and not two_stage and not tridiag" > gpurun_out/r4_small_pytest.log 2>&1; tail -2 gpurun_out/r4_small_pytest.log
echo "== new (fixed-width rows of block words)"; python scratch/r4_small_time.py 2>&1 | grep "^\["
echo "== old"; BODGE_AMD_LIBRARY=$GRAFT_REPO_ROOT/scratch/ab/libk9old.so python scratch/r4_small_time.py 2>&1 | grep "^\["
echo "== new"; python scratch/r4_small_time.py 2>&1 | grep "^\["
