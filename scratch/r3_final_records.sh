#!/bin/bash
# bench records + profiles of the final code (run after any change to the headline kernels or their schedule)
cd $GRAFT_REPO_ROOT
bash scratch/r3_profiles.sh > gpurun_out/r3p_log.txt 2>&1
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/r3_bench_default_final.json 2> gpurun_out/r3_bench_default_final.err &&
python bench.py --steps 20 --warmup 5 > gpurun_out/r3_bench_s20_final.json 2> gpurun_out/r3_bench_s20_final.err &&
python - <<'PY'
import json
for f in ("r3_bench_default_final", "r3_bench_s20_final"):
    r = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1]); ro = r["roofline"]
    print(f, round(r["value"]), "frac", round(ro["frac"],4), "launch_ms", round(ro["launch_ms"],5), "window", round(ro["window_ms"],4), "grid", ro["grid"], "eff", round(ro["effective_GBps"]), "traffic", ro["traffic"], "cpu", round(r["cpu_baseline"]["value"]), r["cpu_baseline"]["cores"])
    for k in ("two_step_kernels","one_step_kernels","streamed_blocks_kernels","complex128_kernels","streamed_bonds_kernels","complex128_sweep_kernels"):
        if r.get(k): print("   ", k, round(r[k]["value"]), round(r[k]["frac"],3), r[k]["kernel"], r[k].get("streams"))
PY
grep "cheb_sweep3<bdg::RealPHMode, 2" gpurun_out/r3p/kernel_stats_bench_default.csv | cut -c1-150
