#!/bin/bash
# parity of the streamed forms (2 / 4 lanes, complex site records), then their rates against round 3's shapes
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4os2; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "streamed or chunk_in_one_launch" > $OUT/pytest.log 2>&1; rc=$?
tail -5 $OUT/pytest.log
[ $rc -ne 0 ] && exit $rc
python scratch/kbench.py "os2=" "os4=BODGE_AMD_SWEEP_LANES=4" --model texture --kind z4 --vectors 8 --steps 63 > $OUT/texture.log 2>&1
python scratch/kbench.py "os2=" "os4=BODGE_AMD_SWEEP_LANES=4" --model potential --vectors 8 --steps 63 > $OUT/potential.log 2>&1
python scratch/kbench.py "bonds=" "one_step=BODGE_AMD_SWEEP=0" --model landau --kind z4 --vectors 8 --steps 63 > $OUT/landau.log 2>&1
python scratch/kbench.py "bonds=" --model ssd --vectors 8 --steps 63 > $OUT/ssd.log 2>&1
grep -h "^os\|^bonds\|^one_step" $OUT/texture.log $OUT/potential.log $OUT/landau.log $OUT/ssd.log
