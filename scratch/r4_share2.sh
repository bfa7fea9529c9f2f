#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4buf; mkdir -p $OUT; : > $OUT/share2.log
for st in 20 63; do
python scratch/kbench.py "texture_$st=" "texture_noshare_$st=BODGE_AMD_STREAMED_SHARE=0" --model texture --kind z4 --vectors 8 --steps $st --rounds 4 2>&1 | grep "^texture_" | cut -c1-190 >> $OUT/share2.log
python scratch/kbench.py "potential_$st=" "potential_noshare_$st=BODGE_AMD_STREAMED_SHARE=0" "potential_lanes4_$st=BODGE_AMD_SWEEP_LANES=4" --model potential --vectors 8 --steps $st --rounds 4 2>&1 | grep "^potential_" | cut -c1-190 >> $OUT/share2.log
done
cat $OUT/share2.log
python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --user-calls 0 > $OUT/bench20.json 2>/dev/null
BODGE_AMD_STREAMED_SHARE=0 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --user-calls 0 > $OUT/bench20_noshare.json 2>/dev/null
python - <<'PY'
import json
for f in ('bench20','bench20_noshare'):
    d=json.loads([l for l in open(f'gpurun_out/r4buf/{f}.json') if l.startswith('{')][0])
    print(f, round(d['value']))
    for k in ('streamed_blocks_kernels','complex128_kernels','streamed_bonds_kernels','complex128_bonds_kernels'):
        v=d.get(k); print('  ',k, round(v['value']), v['kernel'], 'streams', v['streams'], 'launch_ms', round(v['launch_ms'],4), 'window', round(v['window_ms'],3))
PY
