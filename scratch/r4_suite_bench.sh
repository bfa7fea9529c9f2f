#!/bin/bash
cd $GRAFT_REPO_ROOT
bash scratch/r4_suite.sh || exit 1
mkdir -p gpurun_out/r4q
python bench.py --steps 20 --warmup 5 > gpurun_out/r4q/bench_n1_steps20_warmup5.json 2> gpurun_out/r4q/bench_n1_steps20_warmup5.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r4q/bench_n1_steps20_warmup5.json') if l.startswith('{')][0])
print('value', d['value'], d['roofline']['frac'])
for k in ('streamed_blocks_kernels','complex128_kernels','streamed_bonds_kernels','complex128_bonds_kernels','complex128_sweep_kernels'):
    v=d.get(k); print(k, v and (round(v['value']), round(v['frac'],3), v['kernel'], v['streams'], round(v['one_step']['value']) if 'one_step' in v else None))
PY
