#!/bin/bash
# does a larger pool of hardware queues (GPU_MAX_HW_QUEUES, default 4) change anything for one process?
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4hwq; mkdir -p $OUT
for rep in 1 2; do
for q in default 8; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > $OUT/bench_$q.$rep.json 2>/dev/null
done; done
unset GPU_MAX_HW_QUEUES
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r4hwq/bench_*.json')):
    d=json.loads([l for l in open(f) if l.startswith('{')][0])
    u=d['user_facing_calls']
    print(f.split('/')[-1], round(d['value']), round(d['roofline']['frac'],3), 'blocks', round(d['streamed_blocks_kernels']['value']), 'c128', round(d['complex128_kernels']['value']), 'bonds', round(d['streamed_bonds_kernels']['value']), 'cbonds', round(d['complex128_bonds_kernels']['value']),
          'fe', round(u['free_energy_512x64_wall_s']['first_call'],3), round(u['free_energy_512x64_wall_s']['repeated'],3), 'diag', {k: round(v['repeated'],3) for k,v in u['diagonalize_wall_s'].items()}, 'ldos', round(u['ldos_wall_s']['repeated'],4))
PY
