#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc3d; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -- python3 $GRAFT_REPO_ROOT/bench.py --model dwave --lattice 100,100,100 --steps 8 --warmup 2 --cpu-seconds 0 > $OUT/pmc_$c.log 2>&1 || { tail $OUT/pmc_$c.log; exit 1; }
done
cd $GRAFT_REPO_ROOT
find $OUT -name "*_agent_info.csv" -delete
python3 tools/pmc_traffic.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE --workload "100x100x100 R=8" --out $OUT/traffic.json | tail -30
