#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "batches_of_one_call or multi_step_sweep or last_launch or unit or recurrence_dots or ldos or free_energy" 2>&1 | tail -3 &&
python scratch/kbench.py "s1=BODGE_AMD_STREAMS=1" "s2=" "s3=BODGE_AMD_STREAMS=3" "s4=BODGE_AMD_STREAMS=4" --vectors 16 --steps 63 --rounds 3 2>&1 | grep "^s[1-4]" | cut -c1-110 &&
python scratch/kbench.py "s1=BODGE_AMD_STREAMS=1" "s2=" --vectors 8 --steps 63 --rounds 3 2>&1 | grep "^s[1-4]" | cut -c1-110 &&
python scratch/kbench.py "s1x4=BODGE_AMD_STREAMS=1,BODGE_AMD_SWEEP_LANES=4" "s2x4=BODGE_AMD_SWEEP_LANES=4" --vectors 16 --steps 63 --rounds 3 2>&1 | grep "^s[1-4]" | cut -c1-110 &&
python scratch/kbench.py "k1_s1=BODGE_AMD_STREAMS=1,BODGE_AMD_SWEEP=0" "k1_s2=BODGE_AMD_SWEEP=0" "k1_s3=BODGE_AMD_SWEEP=0,BODGE_AMD_STREAMS=3" --vectors 16 --steps 63 --rounds 3 2>&1 | grep "^k1" | cut -c1-110 &&
python scratch/kbench.py "d_s1=BODGE_AMD_STREAMS=1" "d_s2=" "d_s3=BODGE_AMD_STREAMS=3" --lattice 100,100,100 --model dwave --vectors 16 --steps 63 --rounds 3 2>&1 | grep "^d_s" | cut -c1-110 &&
python scratch/kbench.py "p_s1=BODGE_AMD_STREAMS=1" "p_s2=" "p_s3=BODGE_AMD_STREAMS=3" --model potential --vectors 16 --steps 63 --rounds 3 2>&1 | grep "^p_s" | cut -c1-110 &&
python scratch/kbench.py "sm_s1=BODGE_AMD_STREAMS=1" "sm_s2=" --lattice 200,200,1 --vectors 256 --steps 63 --rounds 3 2>&1 | grep "^sm_s" | cut -c1-110
