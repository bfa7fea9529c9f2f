#!/bin/bash
cd $GRAFT_REPO_ROOT
python scratch/kbench.py "seg52=" "seg39=BODGE_AMD_SWEEP_SEGMENTS=39" "seg26=BODGE_AMD_SWEEP_SEGMENTS=26" "seg20=BODGE_AMD_SWEEP_SEGMENTS=20" "seg13=BODGE_AMD_SWEEP_SEGMENTS=13" "seg78=BODGE_AMD_SWEEP_SEGMENTS=78" --vectors 8 --steps 63 --rounds 4 2>&1 | grep "^seg" | cut -c1-140
python scratch/kbench.py "l4seg20=BODGE_AMD_SWEEP_LANES=4" "l4seg10=BODGE_AMD_SWEEP_LANES=4,BODGE_AMD_SWEEP_SEGMENTS=10" "l4seg14=BODGE_AMD_SWEEP_LANES=4,BODGE_AMD_SWEEP_SEGMENTS=14" "l4seg30=BODGE_AMD_SWEEP_LANES=4,BODGE_AMD_SWEEP_SEGMENTS=30" --vectors 16 --steps 63 --rounds 3 2>&1 | grep "^l4seg" | cut -c1-140
python scratch/kbench.py "d=" "dseg1=BODGE_AMD_SWEEP_SEGMENTS=1" "dseg3=BODGE_AMD_SWEEP_SEGMENTS=3" "dseg4=BODGE_AMD_SWEEP_SEGMENTS=4" --lattice 100,100,100 --model dwave --vectors 16 --steps 63 --rounds 3 2>&1 | grep "^d" | cut -c1-140
