#!/bin/bash
# cheb_march3 with buffer addressing (no scratch in its unit body any more?) against one cheb_sweep3 launch per sweep
cd $GRAFT_REPO_ROOT
LOG=gpurun_out/r04_march_buffer.log
{
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "chunk_in_one_launch or gives_up_waiting" 2>&1 | tail -2
for v in 4 8 16; do
echo "== 1000x1000 s-wave+Zeeman, $v vectors, 63 steps per call"
timeout -k 10 300 python scratch/kbench.py "sweeps=BODGE_AMD_MARCH=0" "tickets=BODGE_AMD_MARCH=1" "fixed=BODGE_AMD_MARCH=3" "fixed_nowait=BODGE_AMD_MARCH=3,BODGE_AMD_MARCH_DEBUG=2" "grouped=BODGE_AMD_MARCH=2" --vectors $v --steps 63 --rounds 4 2>&1 | grep "^sweeps\|^tickets\|^fixed\|^grouped" | cut -c1-128
done
echo "== 8 vectors, 20 steps per call (the driver's flags)"
timeout -k 10 300 python scratch/kbench.py "sweeps=BODGE_AMD_MARCH=0" "tickets=BODGE_AMD_MARCH=1" "fixed=BODGE_AMD_MARCH=3" "grouped=BODGE_AMD_MARCH=2" --vectors 8 --steps 20 --rounds 5 2>&1 | grep "^sweeps\|^tickets\|^fixed\|^grouped" | cut -c1-128
} > $LOG 2>&1
cat $LOG
