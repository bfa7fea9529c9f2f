#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/gen
python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "multi_step_sweep or dots or reproducib or free_energy or random" 2>&1 | tail -4 | tee gpurun_out/gen/pytest.log
python3 scratch/call_overhead.py | tee gpurun_out/gen/overhead.log
BODGE_AMD_SWEEP_GEN=0 BODGE_AMD_NO_BATCH_PIPELINE=1 python3 scratch/call_overhead.py | tee -a gpurun_out/gen/overhead.log
BODGE_AMD_SWEEP_GEN=0 python3 scratch/call_overhead.py | tee -a gpurun_out/gen/overhead.log
CO_STEPS=256 python3 scratch/call_overhead.py | tee -a gpurun_out/gen/overhead.log
python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 > gpurun_out/gen/bench20.json 2> gpurun_out/gen/bench20.err; tail -c 600 gpurun_out/gen/bench20.err; python3 -c "
import json; d=json.load(open('gpurun_out/gen/bench20.json')); print('bench --steps 20:', d['value'], d['roofline']['launch_ms'], d['roofline']['frac'])"
