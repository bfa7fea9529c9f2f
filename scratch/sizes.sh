#!/bin/bash
for lib in hint_0000 hint_0011; do
  for lat in 350,350,1 500,500,1 700,700,1 1000,1000,1 1400,1400,1; do
    echo "== $lib $lat"
    BODGE_AMD_LIBRARY=$PWD/scratch/_libs/$lib.so timeout -k 10 200 python3 scratch/kbench.py "$lib=" --rounds 4 --lattice $lat | grep -v "^4 " || exit 1
  done
done
