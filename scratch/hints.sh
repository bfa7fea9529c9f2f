#!/bin/bash
# usage: scratch/hints.sh [kbench options]; runs every scratch/_libs/hint_*.so twice (ABAB order)
for pass in 1 2; do
  for lib in scratch/_libs/hint_*.so; do
    name=$(basename $lib .so)
    BODGE_AMD_LIBRARY=$PWD/$lib timeout -k 10 120 python3 scratch/kbench.py "$name=" --rounds 4 "$@" || exit 1
  done
done
