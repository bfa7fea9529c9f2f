#!/bin/bash
cd $GRAFT_REPO_ROOT
python scratch/kbench.py "d8_l4=" "d8_l2_s2=BODGE_AMD_SWEEP_LANES=2" "d8_l2_s1=BODGE_AMD_SWEEP_LANES=2,BODGE_AMD_STREAMS=1" --lattice 100,100,100 --model dwave --vectors 8 --steps 63 --rounds 3 2>&1 | grep "^d8" | cut -c1-130
python scratch/kbench.py "d16_l4_s2=" "d16_l2_s2=BODGE_AMD_SWEEP_LANES=2" "d16_l4_s1=BODGE_AMD_STREAMS=1" --lattice 100,100,100 --model dwave --vectors 16 --steps 63 --rounds 3 2>&1 | grep "^d16" | cut -c1-130
python scratch/kbench.py "d64_l4_s2=" "d64_l2_s2=BODGE_AMD_SWEEP_LANES=2" "d64_l4_s1=BODGE_AMD_STREAMS=1" --lattice 100,100,100 --model dwave --vectors 64 --steps 63 --rounds 2 2>&1 | grep "^d64" | cut -c1-130
