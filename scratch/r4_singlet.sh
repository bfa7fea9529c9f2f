#!/bin/bash
# "singlet" blocks (A diagonal, B and C antidiagonal) multiplied by their eight non-zero entries (product) against full blocks (nosinglet)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4sing; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_physics.py -x -q -m gpu -k "not dense and not eigen and not jacobi and not ladder and not two_stage and not tridiag" > $OUT/pytest.log 2>&1; rc=$?
tail -2 $OUT/pytest.log
[ $rc -ne 0 ] && exit $rc
: > $OUT/summary.log
for rep in 1 2; do
for lib in product nosinglet; do
  [ $lib = product ] && unset BODGE_AMD_LIBRARY || export BODGE_AMD_LIBRARY=$GRAFT_REPO_ROOT/scratch/ab/lib$lib.so
  echo "== $lib" >> $OUT/summary.log
  python scratch/kbench.py "swave8=" --model swave --vectors 8 --steps 63 --rounds 4 2>&1 | grep "^swave8" | cut -c1-150 >> $OUT/summary.log
  python scratch/kbench.py "swave8_20=" --model swave --vectors 8 --steps 20 --rounds 5 2>&1 | grep "^swave8" | cut -c1-150 >> $OUT/summary.log
  python scratch/kbench.py "swave4=" --model swave --vectors 4 --steps 63 --rounds 4 2>&1 | grep "^swave4" | cut -c1-150 >> $OUT/summary.log
  python scratch/kbench.py "swave_z4_4=" --model swave --kind z4 --vectors 4 --steps 63 --rounds 4 2>&1 | grep "^swave_z4" | cut -c1-150 >> $OUT/summary.log
  python scratch/kbench.py "dwave2d_8=" --model dwave --vectors 8 --steps 63 --rounds 4 2>&1 | grep "^dwave2d" | cut -c1-150 >> $OUT/summary.log
  python scratch/kbench.py "peierls4=" --model peierls --kind z4 --vectors 4 --steps 63 --rounds 4 2>&1 | grep "^peierls4" | cut -c1-150 >> $OUT/summary.log
done; done
cat $OUT/summary.log
