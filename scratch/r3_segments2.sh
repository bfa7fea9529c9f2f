#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "multi_step_sweep or three_step or batches or config3 or position_dependent or last_launch or unit" 2>&1 | tail -2 &&
python scratch/kbench.py "auto=" "seg26=BODGE_AMD_SWEEP_SEGMENTS=26" "seg30=BODGE_AMD_SWEEP_SEGMENTS=30" "seg24=BODGE_AMD_SWEEP_SEGMENTS=24" "s3=BODGE_AMD_STREAMS=3" "s4=BODGE_AMD_STREAMS=4" --vectors 16 --steps 63 --rounds 3 2>&1 | grep "^auto\|^s[0-9e]" | cut -c1-150
python scratch/kbench.py "p_auto=" "p_seg20=BODGE_AMD_SWEEP_SEGMENTS=20" --model potential --vectors 16 --steps 63 --rounds 3 2>&1 | grep "^p_" | cut -c1-150
python scratch/kbench.py "c_auto=" "c_s1=BODGE_AMD_STREAMS=1" --kind z4 --vectors 8 --steps 63 --rounds 3 2>&1 | grep "^c_" | cut -c1-150
python bench.py --cpu-seconds 0 > gpurun_out/r3_bench_default_v5.json 2>/dev/null && python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > gpurun_out/r3_bench_s20_v5.json 2>/dev/null
python - <<'PY'
import json
for f in ("r3_bench_default_v5", "r3_bench_s20_v5"):
    r = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1]); ro = r["roofline"]
    print(f, round(r["value"]), "frac", round(ro["frac"],4), "launch_ms", round(ro["launch_ms"],5), "grid", ro["grid"], "eff", round(ro["effective_GBps"]))
PY
