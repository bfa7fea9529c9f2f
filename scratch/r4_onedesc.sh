#!/bin/bash
# one buffer descriptor per plane (product) against one per component and plane (fourdesc)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4one; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sweep or streamed or chunk or stencil or periodic or lattice" > $OUT/pytest.log 2>&1; rc=$?
tail -2 $OUT/pytest.log
[ $rc -ne 0 ] && exit $rc
: > $OUT/summary.log
for rep in 1 2; do
for lib in product fourdesc; do
  [ $lib = product ] && unset BODGE_AMD_LIBRARY || export BODGE_AMD_LIBRARY=$GRAFT_REPO_ROOT/scratch/ab/lib$lib.so
  echo "== $lib" >> $OUT/summary.log
  python scratch/kbench.py "swave8=" --model swave --vectors 8 --steps 63 --rounds 4 2>&1 | grep "^swave8" | cut -c1-150 >> $OUT/summary.log
  python scratch/kbench.py "swave8_20=" --model swave --vectors 8 --steps 20 --rounds 5 2>&1 | grep "^swave8" | cut -c1-150 >> $OUT/summary.log
  python scratch/kbench.py "swave4=" --model swave --vectors 4 --steps 63 --rounds 4 2>&1 | grep "^swave4" | cut -c1-150 >> $OUT/summary.log
  python scratch/kbench.py "peierls4=" --model peierls --kind z4 --vectors 4 --steps 63 --rounds 4 2>&1 | grep "^peierls4" | cut -c1-150 >> $OUT/summary.log
  python scratch/kbench.py "potential8=" --model potential --vectors 8 --steps 63 --rounds 4 2>&1 | grep "^potential8" | cut -c1-150 >> $OUT/summary.log
  python scratch/kbench.py "texture8=" --model texture --kind z4 --vectors 8 --steps 63 --rounds 4 2>&1 | grep "^texture8" | cut -c1-150 >> $OUT/summary.log
  python scratch/kbench.py "landau8=" --model landau --kind z4 --vectors 8 --steps 63 --rounds 4 2>&1 | grep "^landau8" | cut -c1-150 >> $OUT/summary.log
done; done
cat $OUT/summary.log
