#!/bin/bash
# A/B of two builds of the library: scratch/libbase.so vs scratch/libvariant.so, alternating processes
mkdir -p gpurun_out/ab; : > gpurun_out/ab/libs.log
for rep in 1 2 3; do
  BODGE_AMD_LIBRARY=$PWD/scratch/libbase.so python3 scratch/sweep_ab.py base: 2>&1 | sed "s/^/[base    $rep] /"
  BODGE_AMD_LIBRARY=$PWD/scratch/libvariant.so python3 scratch/sweep_ab.py variant: 2>&1 | sed "s/^/[variant $rep] /"
done | tee -a gpurun_out/ab/libs.log
BODGE_AMD_LIBRARY=$PWD/scratch/libvariant.so python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "multi_step_sweep and swave" 2>&1 | tail -3 | tee -a gpurun_out/ab/libs.log
