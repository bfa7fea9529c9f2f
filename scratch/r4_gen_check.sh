#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4gen
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_physics.py -x -q -m gpu -k "not dense and not eigen and not jacobi and not ladder and not two_stage and not tridiag" > gpurun_out/r4gen/pytest.log 2>&1; rc=$?; tail -3 gpurun_out/r4gen/pytest.log; [ $rc -ne 0 ] && exit $rc
python scratch/kbench.py "swave8_20=" --model swave --vectors 8 --steps 20 --rounds 6 2>&1 | grep "^swave8" | cut -c1-190
python scratch/kbench.py "swave8_63=" --model swave --vectors 8 --steps 63 --rounds 4 2>&1 | grep "^swave8" | cut -c1-190
python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --user-calls 0 > gpurun_out/r4gen/bench20.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --user-calls 0 > gpurun_out/r4gen/bench20b.json 2>/dev/null
python - <<'PY'
import json
for f in ('bench20','bench20b'):
    d=json.loads([l for l in open(f'gpurun_out/r4gen/{f}.json') if l.startswith('{')][0])
    print(f, round(d['value']), round(d['roofline']['frac'],4), 'window', round(d['roofline']['window_ms'],4), 'launch_ms', round(d['roofline']['launch_ms'],4))
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4gen/stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --user-calls 0 > /dev/null 2>&1
grep "cheb_sweep3<bdg::RealPHMode, 2, .*, 0, 4" $(find $GRAFT_REPO_ROOT/gpurun_out/r4gen/stats -name "*kernel_stats.csv" | head -1) | cut -c1-140
