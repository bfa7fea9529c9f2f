#!/bin/bash
cd $GRAFT_REPO_ROOT
for L in 200 250 300 350 400; do
 for v in 64 16 8; do
  python scratch/kbench.py "one_step_${L}_$v=BODGE_AMD_SWEEP=0" "sweep_${L}_$v=BODGE_AMD_SWEEP=1" --lattice $L,$L,1 --vectors $v --steps 63 --rounds 3 2>&1 | grep "^one_step\|^sweep" | cut -c1-70
 done
done
