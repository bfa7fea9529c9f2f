#!/bin/bash
# A/B of the on-site record loads (plain vs non-temporal): time and PMC traffic of cheb_sweep3 OS, real and complex
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3os; rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python scratch/kbench.py "plain=" "nt=BODGE_AMD_SWEEP_STREAM=9" --model potential --steps 63 > $OUT/time_real.log 2>&1
python scratch/kbench.py "plain=" "nt=BODGE_AMD_SWEEP_STREAM=9" --model texture --kind z4 --vectors 4 --steps 63 > $OUT/time_complex.log 2>&1
grep -h "^plain\|^nt" $OUT/time_real.log $OUT/time_complex.log
cd /tmp && export TMPDIR=/tmp
for v in plain nt; do
  for c in FETCH_SIZE WRITE_SIZE; do
    for m in potential texture; do
      kind=rademacher; vec=8; [ $m = texture ] && kind=z4 && vec=4
      arg="$v="; [ $v = nt ] && arg="nt=BODGE_AMD_SWEEP_STREAM=9"
      timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_${v}_${m}_$c -- python3 $GRAFT_REPO_ROOT/scratch/kbench.py "$arg" --model $m --kind $kind --vectors $vec --steps 63 --rounds 2 > /dev/null 2> $OUT/pmc_${v}_${m}_$c.err || echo "pmc failed $v $m $c"
    done
  done
done
cd $GRAFT_REPO_ROOT
for v in plain nt; do for m in potential texture; do
  echo "== $v $m"; python3 tools/pmc_traffic.py $OUT/pmc_${v}_${m}_FETCH_SIZE $OUT/pmc_${v}_${m}_WRITE_SIZE --workload "$v $m" --out $OUT/traffic_ab.json > /dev/null
done; done
python3 -c "
import json; t=json.load(open('$OUT/traffic_ab.json'))
for k,v in sorted(t.items()): print(k, round(v['traffic_bytes_per_launch']/1e6,1), 'read', round(v['read_bytes']/1e6,1), 'write', round(v['write_bytes']/1e6,1))
"
rm -rf $OUT/pmc_*_FETCH_SIZE $OUT/pmc_*_WRITE_SIZE
