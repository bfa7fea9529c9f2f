#!/bin/bash
# PMC traffic passes only (see r2_profiles.sh): one counter per pass, full-length chunks so that the median launch is a full one
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2q; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 63 --warmup 3 --cpu-seconds 0 > $OUT/pmc_$c.json 2> $OUT/pmc_$c.err || echo "pmc $c failed"
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_dwave_$c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 32 --warmup 2 --cpu-seconds 0 --model dwave --lattice 100,100,100 > $OUT/pmc_dwave_$c.json 2> $OUT/pmc_dwave_$c.err || echo "pmc dwave $c failed"
done
cd $GRAFT_REPO_ROOT
cp profiles/traffic.json $OUT/traffic.json
python3 tools/pmc_traffic.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE --workload "1000x1000x1 R=8" --out $OUT/traffic.json > /dev/null
python3 tools/pmc_traffic.py $OUT/pmc_dwave_FETCH_SIZE $OUT/pmc_dwave_WRITE_SIZE --workload "100x100x100 R=8" --out $OUT/traffic.json > /dev/null
mkdir -p $OUT/pmc; for d in pmc_FETCH_SIZE pmc_WRITE_SIZE pmc_dwave_FETCH_SIZE pmc_dwave_WRITE_SIZE; do cp $(find $OUT/$d -name "*counter_collection.csv" | head -1) $OUT/pmc/${d#pmc_}_counter_collection.csv; done
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_dwave_FETCH_SIZE $OUT/pmc_dwave_WRITE_SIZE
