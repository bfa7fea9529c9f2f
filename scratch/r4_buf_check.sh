#!/bin/bash
# buffer addressing in the three-step sweep: parity of everything that sweeps, then rates (product = buffer ops everywhere,
# flat = BDG_SWEEP_BUFFER_OPS=0: flat addressing in the dictionary forms)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4buf; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sweep or streamed or chunk or stencil or periodic or lattice" > $OUT/pytest.log 2>&1; rc=$?
tail -3 $OUT/pytest.log
[ $rc -ne 0 ] && exit $rc
: > $OUT/summary.log
for lib in product flat; do
  [ $lib = product ] && unset BODGE_AMD_LIBRARY || export BODGE_AMD_LIBRARY=$GRAFT_REPO_ROOT/scratch/ab/lib$lib.so
  echo "== $lib" >> $OUT/summary.log
  python scratch/kbench.py "swave8=" "swave8_lanes4=BODGE_AMD_SWEEP_LANES=4" --model swave --vectors 8 --steps 63 --rounds 4 2>&1 | grep "^swave" | cut -c1-190 >> $OUT/summary.log
  python scratch/kbench.py "swave4=" --model swave --vectors 4 --steps 63 --rounds 4 2>&1 | grep "^swave" | cut -c1-190 >> $OUT/summary.log
  python scratch/kbench.py "peierls4=" --model peierls --kind z4 --vectors 4 --steps 63 --rounds 4 2>&1 | grep "^peierls" | cut -c1-190 >> $OUT/summary.log
done
unset BODGE_AMD_LIBRARY
echo "== streamed forms (product)" >> $OUT/summary.log
python scratch/kbench.py "texture_os2=" "texture_os4=BODGE_AMD_SWEEP_LANES=4" --model texture --kind z4 --vectors 8 --steps 63 --rounds 4 2>&1 | grep "^texture" | cut -c1-190 >> $OUT/summary.log
python scratch/kbench.py "texture_os2_alone=" --model texture --kind z4 --vectors 2 --steps 63 --rounds 4 2>&1 | grep "^texture" | cut -c1-190 >> $OUT/summary.log
python scratch/kbench.py "texture_os4_alone=BODGE_AMD_SWEEP_LANES=4" --model texture --kind z4 --vectors 4 --steps 63 --rounds 4 2>&1 | grep "^texture" | cut -c1-190 >> $OUT/summary.log
python scratch/kbench.py "potential_os2=" "potential_os4=BODGE_AMD_SWEEP_LANES=4" --model potential --vectors 8 --steps 63 --rounds 4 2>&1 | grep "^potential" | cut -c1-190 >> $OUT/summary.log
python scratch/kbench.py "landau=" --model landau --kind z4 --vectors 8 --steps 63 --rounds 4 2>&1 | grep "^landau" | cut -c1-190 >> $OUT/summary.log
python scratch/kbench.py "ssd=" --model ssd --vectors 8 --steps 63 --rounds 4 2>&1 | grep "^ssd" | cut -c1-190 >> $OUT/summary.log
cat $OUT/summary.log
