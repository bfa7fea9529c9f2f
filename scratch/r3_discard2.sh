#!/bin/bash
cd $GRAFT_REPO_ROOT
bash scratch/r3_profiles.sh > gpurun_out/r3p_log.txt 2>&1
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/r3_bench_default_v3.json 2> gpurun_out/r3_bench_default_v3.err &&
python bench.py --steps 20 --warmup 5 > gpurun_out/r3_bench_s20_v3.json 2> gpurun_out/r3_bench_s20_v3.err &&
BODGE_AMD_KEEP_LAST=1 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > gpurun_out/r3_bench_s20_keep.json 2>/dev/null &&
python - <<'PY'
import json
for f in ("r3_bench_default_v3", "r3_bench_s20_v3", "r3_bench_s20_keep"):
    r = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
    ro = r["roofline"]
    print(f, round(r["value"]), round(ro["frac"],4), round(ro["launch_ms"],5), round(ro["bytes_per_launch"]/1e6,1), round(ro["effective_GBps"]), ro["traffic"])
PY
tail -5 gpurun_out/r3p_log.txt
