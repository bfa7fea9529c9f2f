#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4buf; mkdir -p $OUT
python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --user-calls 0 > $OUT/bench20.json 2>/dev/null
BODGE_AMD_STREAMED_SHARE=0 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --user-calls 0 > $OUT/bench20_noshare.json 2>/dev/null
python bench.py --cpu-seconds 0 --user-calls 0 > $OUT/bench256.json 2>/dev/null
python - <<'PY'
import json
for f in ('bench20','bench20_noshare','bench256'):
    d=json.loads([l for l in open(f'gpurun_out/r4buf/{f}.json') if l.startswith('{')][0])
    print(f, round(d['value']), d['roofline']['frac'])
    for k in ('streamed_blocks_kernels','complex128_kernels','streamed_bonds_kernels','complex128_bonds_kernels','complex128_sweep_kernels','two_step_kernels','one_step_kernels'):
        v=d.get(k); print('  ',k, round(v['value']), round(v['frac'],3), v['kernel'], 'streams', v['streams'], 'launch_ms', round(v['launch_ms'],4), 'window', round(v['window_ms'],3))
PY
