#!/bin/bash
for lat in 1000,1000,1 1400,1400,1; do
  echo "== $lat"
  timeout -k 10 200 python3 scratch/kbench.py "s0=BODGE_AMD_STREAM_VECTORS=0" "s1_prev=BODGE_AMD_STREAM_VECTORS=1" "s2_store=BODGE_AMD_STREAM_VECTORS=2" "s3_both=BODGE_AMD_STREAM_VECTORS=3" --lattice $lat || exit 1
done
