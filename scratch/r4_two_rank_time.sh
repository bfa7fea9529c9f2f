#!/bin/bash
# where the two-rank rehearsal of bench.py on one GPU spends its time
cd $GRAFT_REPO_ROOT
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 WORLD_SIZE=2 LOCAL_WORLD_SIZE=2 TORCHELASTIC_RUN_ID=rehearsal BODGE_AMD_TRACE_BENCH=1
date +%T.%N
for r in 0 1; do
  RANK=$r LOCAL_RANK=$r timeout -k 10 400 python bench.py --gpus 2 --allow-gloo --lattice 64,64,1 --steps 12 --warmup 3 --cpu-seconds 0 > /tmp/rank$r.out 2> /tmp/rank$r.err &
done
wait
date +%T.%N
python - <<'PY'
import json
line=[l for l in open('/tmp/rank0.out') if l.startswith('{')][0]
rec=json.loads(line)
print(rec["config"]["rccl_load_s"], rec["config"]["collective"][:200])
PY
tail -5 /tmp/rank0.err /tmp/rank1.err
