#!/bin/bash
# the chunk kernel with ONE unit body (no start-block generation inside: scratch 0) against one launch per sweep
cd $GRAFT_REPO_ROOT
LOG=gpurun_out/r04_march_one_body.log
export BODGE_AMD_LIBRARY=$GRAFT_REPO_ROOT/scratch/ab/libmarch.so
{
for v in 4 8 16; do
echo "== 1000x1000 s-wave+Zeeman, $v vectors, 63 steps per call"
timeout -k 10 300 python scratch/kbench.py "sweeps=BODGE_AMD_MARCH=0" "sweeps_fill=BODGE_AMD_MARCH=0,BODGE_AMD_SWEEP_GEN=0" "tickets=BODGE_AMD_MARCH=1,BODGE_AMD_SWEEP_GEN=0" "fixed=BODGE_AMD_MARCH=3,BODGE_AMD_SWEEP_GEN=0" "fixed_nowait=BODGE_AMD_MARCH=3,BODGE_AMD_MARCH_DEBUG=2,BODGE_AMD_SWEEP_GEN=0" --vectors $v --steps 63 --rounds 4 2>&1 | grep "^sweeps\|^tickets\|^fixed" | cut -c1-128
done
echo "== 8 vectors, 20 steps per call (the driver's flags)"
timeout -k 10 300 python scratch/kbench.py "sweeps=BODGE_AMD_MARCH=0" "tickets=BODGE_AMD_MARCH=1,BODGE_AMD_SWEEP_GEN=0" "fixed=BODGE_AMD_MARCH=3,BODGE_AMD_SWEEP_GEN=0" --vectors 8 --steps 20 --rounds 5 2>&1 | grep "^sweeps\|^tickets\|^fixed" | cut -c1-128
echo "== 8 vectors, 256 steps per call"
timeout -k 10 300 python scratch/kbench.py "sweeps=BODGE_AMD_MARCH=0" "tickets=BODGE_AMD_MARCH=1,BODGE_AMD_SWEEP_GEN=0" "fixed=BODGE_AMD_MARCH=3,BODGE_AMD_SWEEP_GEN=0" --vectors 8 --steps 256 --rounds 3 2>&1 | grep "^sweeps\|^tickets\|^fixed" | cut -c1-128
} > $LOG 2>&1
cat $LOG
