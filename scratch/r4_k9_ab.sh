#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "== new"; python scratch/r4_k9_time.py 2>&1 | grep "^L="
echo "== old"; BODGE_AMD_LIBRARY=$GRAFT_REPO_ROOT/scratch/ab/libk9old.so python scratch/r4_k9_time.py 2>&1 | grep "^L="
echo "== new"; python scratch/r4_k9_time.py 2>&1 | grep "^L="
