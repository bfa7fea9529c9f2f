#!/bin/bash
timeout -k 10 300 python3 scratch/kbench.py "auto=" "cu2=BODGE_AMD_BLOCKS_PER_CU=2" "cu3=BODGE_AMD_BLOCKS_PER_CU=3" "cu4=BODGE_AMD_BLOCKS_PER_CU=4" "cu5=BODGE_AMD_BLOCKS_PER_CU=5" "cu6=BODGE_AMD_BLOCKS_PER_CU=6" "cu8=BODGE_AMD_BLOCKS_PER_CU=8" --rounds 4
