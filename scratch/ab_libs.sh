#!/bin/bash
# A/B of two builds of the library: scratch/libbase.so (the earlier build) vs the in-tree product, alternating processes
mkdir -p gpurun_out/ab
python3 -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "multi_step_sweep" 2>&1 | tail -3 | tee gpurun_out/ab/libs.log
for rep in 1 2 3; do
  BODGE_AMD_LIBRARY=$PWD/scratch/libbase.so python3 scratch/sweep_ab.py base: 2>&1 | sed "s/^/[base $rep] /"
  python3 scratch/sweep_ab.py new: 2>&1 | sed "s/^/[new  $rep] /"
done | tee -a gpurun_out/ab/libs.log
AB_LATTICE=700,700,1 BODGE_AMD_LIBRARY=$PWD/scratch/libbase.so python3 scratch/sweep_ab.py base700: | tee -a gpurun_out/ab/libs.log
AB_LATTICE=700,700,1 python3 scratch/sweep_ab.py new700: | tee -a gpurun_out/ab/libs.log
