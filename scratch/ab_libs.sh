#!/bin/bash
# A/B of two builds of the library (product vs scratch/libvariant.so), alternating processes
mkdir -p gpurun_out/ab
for rep in 1 2 3; do
  python scratch/sweep_ab.py base: 2>&1 | sed "s/^/[product $rep] /"
  BODGE_AMD_LIBRARY=$PWD/scratch/libvariant.so python scratch/sweep_ab.py variant: 2>&1 | sed "s/^/[variant $rep] /"
done | tee gpurun_out/ab/libs.log
BODGE_AMD_LIBRARY=$PWD/scratch/libvariant.so python -m pytest tests/test_gpu_parity.py -q -m gpu -k "multi_step_sweep" 2>&1 | tail -3 | tee -a gpurun_out/ab/libs.log
