#!/bin/bash
# K8: units cut for equal length (default) against whole-column segments (BODGE_AMD_ROLL_CHUNKS=0)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4chunks; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py -x -q -m gpu -k "roll or 3d or cubed or three_d or slab or dwave or lattice or stencil" > $OUT/pytest.log 2>&1; rc=$?
tail -2 $OUT/pytest.log
[ $rc -ne 0 ] && exit $rc
: > $OUT/summary.log
for v in 8 16 4; do
  python scratch/kbench.py "chunks_$v=" "segments_$v=BODGE_AMD_ROLL_CHUNKS=0" --lattice 100,100,100 --model dwave --vectors $v --steps 63 --rounds 4 2>&1 | grep "^chunks\|^segments" | cut -c1-200 >> $OUT/summary.log
done
python scratch/kbench.py "chunks_z4_8=" "segments_z4_8=BODGE_AMD_ROLL_CHUNKS=0" --lattice 100,100,100 --model dwave --kind z4 --vectors 8 --steps 63 --rounds 4 2>&1 | grep "^chunks\|^segments" | cut -c1-200 >> $OUT/summary.log
python scratch/kbench.py "chunks_64cube_8=" "segments_64cube_8=BODGE_AMD_ROLL_CHUNKS=0" --lattice 128,96,96 --model dwave --vectors 8 --steps 63 --rounds 4 2>&1 | grep "^chunks\|^segments" | cut -c1-200 >> $OUT/summary.log
cat $OUT/summary.log
