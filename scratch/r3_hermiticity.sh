#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3h; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $GRAFT_REPO_ROOT/scratch/hermiticity_probe.py > $OUT/time.log 2>&1; cat $OUT/time.log
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -- python3 $GRAFT_REPO_ROOT/scratch/hermiticity_probe.py > /dev/null 2> $OUT/pmc_$c.err || echo "pmc $c failed"
done
cd $GRAFT_REPO_ROOT
python3 tools/pmc_traffic.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE --workload "1000x1000x1" --out $OUT/traffic.json | grep -A8 hermiticity
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
