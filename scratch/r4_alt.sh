#!/bin/bash
cd $GRAFT_REPO_ROOT
python scratch/kbench.py "texture=" "texture_noalt=BODGE_AMD_ALTERNATE=0" "texture_nozig=BODGE_AMD_SWEEP_ZIGZAG=0" "texture_neither=BODGE_AMD_ALTERNATE=0,BODGE_AMD_SWEEP_ZIGZAG=0" --model texture --kind z4 --vectors 8 --steps 63 --rounds 4 2>&1 | grep "^texture" | cut -c1-110
python scratch/kbench.py "cswave=" "cswave_noalt=BODGE_AMD_ALTERNATE=0" "cswave_nozig=BODGE_AMD_SWEEP_ZIGZAG=0" --model swave --kind z4 --vectors 8 --steps 63 --rounds 4 2>&1 | grep "^cswave" | cut -c1-110
python scratch/kbench.py "swave=" "swave_noalt=BODGE_AMD_ALTERNATE=0" "swave_nozig=BODGE_AMD_SWEEP_ZIGZAG=0" --model swave --vectors 8 --steps 63 --rounds 4 2>&1 | grep "^swave" | cut -c1-110
