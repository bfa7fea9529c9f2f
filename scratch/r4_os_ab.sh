#!/bin/bash
# what the streamed on-site records cost: timing builds without the record loads (1), without loads and ring writes (3),
# without the on-site product (4), without all three (7); one lane group alone and the 8-vector call
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4osab; mkdir -p $OUT; : > $OUT/summary.log
for lib in product exp1 exp3 exp4 exp7; do
  [ $lib = product ] && unset BODGE_AMD_LIBRARY || export BODGE_AMD_LIBRARY=$GRAFT_REPO_ROOT/scratch/ab/lib$lib.so
  for cfg in "texture z4 4 4" "texture z4 2 2" "texture z4 8 4" "potential rademacher 8 4" "potential rademacher 4 2"; do
    set -- $cfg
    echo "== $lib $cfg" >> $OUT/summary.log
    python scratch/kbench.py "lanes$4=BODGE_AMD_SWEEP_LANES=$4" --model $1 --kind $2 --vectors $3 --steps 63 --rounds 3 2>&1 | grep "^lanes" | cut -c1-200 >> $OUT/summary.log
  done
done
echo "== dictionary forms alone" >> $OUT/summary.log
unset BODGE_AMD_LIBRARY
python scratch/kbench.py "dict2=" --model peierls --kind z4 --vectors 2 --steps 63 --rounds 3 2>&1 | grep "^dict" | cut -c1-200 >> $OUT/summary.log
python scratch/kbench.py "dict4=BODGE_AMD_SWEEP_LANES=4" --model peierls --kind z4 --vectors 4 --steps 63 --rounds 3 2>&1 | grep "^dict" | cut -c1-200 >> $OUT/summary.log
python scratch/kbench.py "dict2=" --model swave --vectors 4 --steps 63 --rounds 3 2>&1 | grep "^dict" | cut -c1-200 >> $OUT/summary.log
python scratch/kbench.py "dict4=BODGE_AMD_SWEEP_LANES=4" --model swave --vectors 8 --steps 63 --rounds 3 2>&1 | grep "^dict" | cut -c1-200 >> $OUT/summary.log
cat $OUT/summary.log
