#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4q
( while true; do sleep 60; echo "[suite] $(date +%T) $(tail -c 200 gpurun_out/r4q/pytest_rest.log | tr '\n' ' ' | tail -c 120)"; done ) &
TICK=$!
timeout -k 10 1000 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_gpu_physics.py tests/test_gpu_zz_big_libraries.py -x -q -m gpu --durations=25 > gpurun_out/r4q/pytest_rest.log 2>&1
RC=$?
kill $TICK
tail -40 gpurun_out/r4q/pytest_rest.log
[ $RC -ne 0 ] && exit $RC
python bench.py --steps 20 --warmup 5 > gpurun_out/r4q/bench_n1_steps20_warmup5.json 2> gpurun_out/r4q/bench_n1_steps20_warmup5.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r4q/bench_n1_steps20_warmup5.json') if l.startswith('{')][0])
print('value', d['value'], d['roofline']['frac'])
for k in ('streamed_blocks_kernels','complex128_kernels','streamed_bonds_kernels','complex128_bonds_kernels','complex128_sweep_kernels'):
    v=d.get(k); print(k, v and (round(v['value']), round(v['frac'],3), v['kernel'], v['streams'], round(v['one_step']['value']) if 'one_step' in v else None))
PY
