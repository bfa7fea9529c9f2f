#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4buf; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "streamed or chunk" > $OUT/pytest2.log 2>&1; rc=$?
tail -3 $OUT/pytest2.log
[ $rc -ne 0 ] && exit $rc
: > $OUT/summary2.log
python scratch/kbench.py "swave8=" --model swave --vectors 8 --steps 63 --rounds 4 2>&1 | grep "^swave8" | cut -c1-190 >> $OUT/summary2.log
python scratch/kbench.py "texture_os4=" "texture_os2=BODGE_AMD_SWEEP_LANES=2" --model texture --kind z4 --vectors 8 --steps 63 --rounds 4 2>&1 | grep "^texture_" | cut -c1-190 >> $OUT/summary2.log
python scratch/kbench.py "texture_os4_alone=" --model texture --kind z4 --vectors 4 --steps 63 --rounds 4 2>&1 | grep "^texture_" | cut -c1-190 >> $OUT/summary2.log
python scratch/kbench.py "potential_os2=" "potential_os4=BODGE_AMD_SWEEP_LANES=4" --model potential --vectors 8 --steps 63 --rounds 4 2>&1 | grep "^potential_" | cut -c1-190 >> $OUT/summary2.log
python scratch/kbench.py "potential16_os2=" "potential16_os4=BODGE_AMD_SWEEP_LANES=4" --model potential --vectors 16 --steps 63 --rounds 4 2>&1 | grep "^potential" | cut -c1-190 >> $OUT/summary2.log
python scratch/kbench.py "landau=" --model landau --kind z4 --vectors 8 --steps 63 --rounds 4 2>&1 | grep "^landau" | cut -c1-190 >> $OUT/summary2.log
python scratch/kbench.py "ssd=" --model ssd --vectors 8 --steps 63 --rounds 4 2>&1 | grep "^ssd" | cut -c1-190 >> $OUT/summary2.log
cat $OUT/summary2.log
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "position_dependent" > $OUT/pytest3.log 2>&1; tail -3 $OUT/pytest3.log
