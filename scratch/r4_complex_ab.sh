#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for lib in product nosinglet; do
  [ $lib = product ] && unset BODGE_AMD_LIBRARY || export BODGE_AMD_LIBRARY=$GRAFT_REPO_ROOT/scratch/ab/lib$lib.so
  echo "== $lib"
  python scratch/kbench.py "texture8=" --model texture --kind z4 --vectors 8 --steps 63 --rounds 4 2>&1 | grep "^texture8" | cut -c1-150
  python scratch/kbench.py "swave_z4_8=" --model swave --kind z4 --vectors 8 --steps 63 --rounds 4 2>&1 | grep "^swave_z4" | cut -c1-150
  python scratch/kbench.py "landau8=" --model landau --kind z4 --vectors 8 --steps 63 --rounds 4 2>&1 | grep "^landau8" | cut -c1-150
  python scratch/kbench.py "swave8=" --model swave --vectors 8 --steps 63 --rounds 4 2>&1 | grep "^swave8" | cut -c1-150
done; done
