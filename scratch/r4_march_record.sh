#!/bin/bash
# round 4: the persistent chunk kernel cheb_march3 against one cheb_sweep3 launch per sweep.
# parity tests, then interleaved A/B timings with the kernel's own wait / claim statistics (debug bit 3).
# BODGE_AMD_MARCH: 0 classic (two streams), 1 tickets, 3 fixed unit per wave, 2 one launch per sweep for all lane groups
# BODGE_AMD_MARCH_DEBUG: 1 no acquire, 2 no wait (WRONG results: the price of the hand-over), 4 plain stores, 8 statistics
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
LOG=gpurun_out/r04_march.log
{
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "chunk_in_one_launch or gives_up_waiting" 2>&1 | tail -3
for v in 4 8 16; do
echo "== 1000x1000 s-wave+Zeeman, $v vectors, 63 steps per call"
timeout -k 10 300 python scratch/kbench.py "sweeps=BODGE_AMD_MARCH=0" "tickets=BODGE_AMD_MARCH=1,BODGE_AMD_MARCH_DEBUG=8" "tickets_noacq=BODGE_AMD_MARCH=1,BODGE_AMD_MARCH_DEBUG=9" "tickets_nowait=BODGE_AMD_MARCH=1,BODGE_AMD_MARCH_DEBUG=11" "fixed=BODGE_AMD_MARCH=3,BODGE_AMD_MARCH_DEBUG=8" "fixed_noacq=BODGE_AMD_MARCH=3,BODGE_AMD_MARCH_DEBUG=9" "fixed_plain_stores=BODGE_AMD_MARCH=3,BODGE_AMD_MARCH_DEBUG=4" "fixed_sleep4=BODGE_AMD_MARCH=3,BODGE_AMD_MARCH_SLEEP=4" "fixed_sleep16=BODGE_AMD_MARCH=3,BODGE_AMD_MARCH_SLEEP=16" "grouped=BODGE_AMD_MARCH=2" --vectors $v --steps 63 --rounds 3 2>&1 | grep "^sweeps\|^tickets\|^fixed\|^grouped\|bdg\]" | cut -c1-128 | sort | uniq -c | sort -k2
done
echo "== 8 vectors, 20 steps per call (the driver's flags)"
timeout -k 10 300 python scratch/kbench.py "sweeps=BODGE_AMD_MARCH=0" "tickets=BODGE_AMD_MARCH=1" "fixed=BODGE_AMD_MARCH=3" "grouped=BODGE_AMD_MARCH=2" --vectors 8 --steps 20 --rounds 5 2>&1 | grep "^sweeps\|^tickets\|^fixed\|^grouped" | cut -c1-128
echo "== random on-site potential (K7b-OS), 8 and 16 vectors, 63 steps"
for v in 8 16; do
timeout -k 10 300 python scratch/kbench.py "sweeps=BODGE_AMD_MARCH=0" "tickets=BODGE_AMD_MARCH=1" "fixed=BODGE_AMD_MARCH=3" "grouped=BODGE_AMD_MARCH=2" --model potential --vectors $v --steps 63 --rounds 3 2>&1 | grep "^sweeps\|^tickets\|^fixed\|^grouped" | cut -c1-128
done
} > $LOG 2>&1
cat $LOG
