#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -3 &&
bash scratch/r3_profiles.sh > gpurun_out/r3p_log.txt 2>&1
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/r3_bench_default_v4.json 2> gpurun_out/r3_bench_default_v4.err &&
python bench.py --steps 20 --warmup 5 > gpurun_out/r3_bench_s20_v4.json 2> gpurun_out/r3_bench_s20_v4.err &&
python bench.py --lattice 100,100,100 --model dwave > gpurun_out/r3_bench_dwave100_v4.json 2>/dev/null &&
python - <<'PY'
import json
for f in ("r3_bench_default_v4", "r3_bench_s20_v4", "r3_bench_dwave100_v4"):
    r = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
    ro = r["roofline"]
    print(f, round(r["value"]), "frac", round(ro["frac"],4), "launch_ms", round(ro["launch_ms"],5), "window", round(ro["window_ms"],4), "streams", ro["streams"], "launches", ro["launches"], round(ro["bytes_per_launch"]/1e6,1), "eff", round(ro["effective_GBps"]), ro["traffic"])
    for k in ("two_step_kernels","one_step_kernels","streamed_blocks_kernels","complex128_kernels","streamed_bonds_kernels","complex128_sweep_kernels"):
        if r.get(k): print("   ", k, round(r[k]["value"]), round(r[k]["frac"],3), r[k]["kernel"], r[k].get("streams"))
PY
head -4 gpurun_out/r3p/kernel_stats_bench_default.csv | cut -c1-160
