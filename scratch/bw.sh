#!/bin/bash
for v in 16 64; do
  echo "== 1000^2 vectors $v"; timeout -k 10 300 python3 scratch/kbench.py "auto=" "batch64=BODGE_AMD_BATCH=64" --vectors $v --rounds 3 || exit 1
done
echo "== 400^2 vectors 64"; timeout -k 10 300 python3 scratch/kbench.py "auto=" "batch64=BODGE_AMD_BATCH=64" "batch32=BODGE_AMD_BATCH=32" --vectors 64 --rounds 3 --lattice 400,400,1 || exit 1
echo "== 600^2 vectors 64"; timeout -k 10 300 python3 scratch/kbench.py "auto=" "batch64=BODGE_AMD_BATCH=64" "batch16=BODGE_AMD_BATCH=16" "batch8=BODGE_AMD_BATCH=8" --vectors 64 --rounds 3 --lattice 600,600,1 || exit 1
