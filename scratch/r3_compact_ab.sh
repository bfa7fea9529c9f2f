#!/bin/bash
# A/B on one box, alternating processes: compact diagonal copies (product) against diagonal blocks read from the full table entries
cd $GRAFT_REPO_ROOT
for round in 1 2 3; do
  for v in compact nocompact; do
    lib=$GRAFT_REPO_ROOT/bodge_amd/csrc/libbodge_hip.so; [ $v = nocompact ] && lib=$GRAFT_REPO_ROOT/scratch/libbodge_hip_nocompact.so
    BODGE_AMD_LIBRARY=$lib python scratch/kbench.py "$v=" "${v}4=BODGE_AMD_SWEEP_LANES=4" --steps 64 --rounds 4 2>&1 | grep "^$v"
  done
done
