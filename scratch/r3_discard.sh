#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "last_launch or multi_step_sweep or three_step or batches_of_one_call or slab" 2>&1 | tail -3 &&
python bench.py --steps 20 --warmup 5 > gpurun_out/r3_bench_discard_s20.json 2> gpurun_out/r3_bench_discard_s20.err &&
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r3_bench_discard_s20.json").read().strip().splitlines()[-1])
print("s20:", r["value"], r["roofline"]["frac"], r["roofline"]["launch_ms"], r["roofline"]["bytes_per_launch"], r["roofline"]["bytes_full_launch"])
for k in ("two_step_kernels","one_step_kernels","streamed_blocks_kernels","complex128_kernels","streamed_bonds_kernels","complex128_sweep_kernels"):
    if r.get(k): print(k, round(r[k]["value"]), round(r[k]["frac"],3))
PY
