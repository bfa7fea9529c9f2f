#!/bin/bash
V="old=BODGE_AMD_STREAM_VECTORS=0,BODGE_AMD_ALTERNATE=0 alt=BODGE_AMD_STREAM_VECTORS=0,BODGE_AMD_ALTERNATE=1 stream_alt=BODGE_AMD_STREAM_VECTORS=3,BODGE_AMD_ALTERNATE=1 stream=BODGE_AMD_STREAM_VECTORS=3,BODGE_AMD_ALTERNATE=0"
echo "== streamed real PH"; BODGE_AMD_DICT=0 timeout -k 10 200 python3 scratch/kbench.py $V || exit 1
echo "== streamed complex PH"; BODGE_AMD_DICT=0 BODGE_AMD_REAL=0 timeout -k 10 200 python3 scratch/kbench.py $V || exit 1
echo "== dict complex"; BODGE_AMD_REAL=0 timeout -k 10 200 python3 scratch/kbench.py $V || exit 1
