#!/bin/bash
# K8 (3-D rolling kernel) with branch-free buffer addressing (product) against round 3's flat addressing (scratch/ab/libflat.so)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4roll; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "roll or 3d or cubed or three_d or slab or dwave or lattice" > $OUT/pytest.log 2>&1; rc=$?
tail -3 $OUT/pytest.log
[ $rc -ne 0 ] && exit $rc
: > $OUT/summary.log
for rep in 1 2; do
for lib in product flat; do
  [ $lib = product ] && unset BODGE_AMD_LIBRARY || export BODGE_AMD_LIBRARY=$GRAFT_REPO_ROOT/scratch/ab/lib$lib.so
  echo "== $lib" >> $OUT/summary.log
  python scratch/kbench.py "dwave100_8=" --lattice 100,100,100 --model dwave --vectors 8 --steps 63 --rounds 4 2>&1 | grep "^dwave100" | cut -c1-190 >> $OUT/summary.log
  python scratch/kbench.py "dwave100_16=" --lattice 100,100,100 --model dwave --vectors 16 --steps 63 --rounds 4 2>&1 | grep "^dwave100" | cut -c1-190 >> $OUT/summary.log
  python scratch/kbench.py "dwave100_4=" --lattice 100,100,100 --model dwave --vectors 4 --steps 63 --rounds 4 2>&1 | grep "^dwave100" | cut -c1-190 >> $OUT/summary.log
  python scratch/kbench.py "dwave100_z4_8=" --lattice 100,100,100 --model dwave --kind z4 --vectors 8 --steps 63 --rounds 4 2>&1 | grep "^dwave100" | cut -c1-190 >> $OUT/summary.log
done; done
cat $OUT/summary.log
