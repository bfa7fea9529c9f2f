#!/bin/bash
# Round-3 profiles of the 3-D configuration (100^3 d-wave, 8 vectors): rocprofv3 kernel stats of the bench, one PMC pass per counter.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3pd; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/bench.py --model dwave --lattice 100,100,100 --steps 256 --warmup 8 --cpu-seconds 0 > $OUT/bench_dwave100_under_rocprof.json 2> $OUT/stats.err || echo "stats failed"
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_bench_dwave100.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -- python3 $GRAFT_REPO_ROOT/bench.py --model dwave --lattice 100,100,100 --steps 32 --warmup 2 --cpu-seconds 0 > $OUT/pmc_$c.json 2> $OUT/pmc_$c.err || echo "pmc $c failed"
done
cd $GRAFT_REPO_ROOT
cp profiles/traffic.json $OUT/traffic.json
python3 tools/pmc_traffic.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE --workload "100x100x100 R=8" --out $OUT/traffic.json > /dev/null
mkdir -p $OUT/pmc; for d in pmc_FETCH_SIZE pmc_WRITE_SIZE; do cp $(find $OUT/$d -name "*counter_collection.csv" | head -1) $OUT/pmc/dwave_${d#pmc_}_counter_collection.csv; done
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/stats
python3 bench.py --model dwave --lattice 100,100,100 > $OUT/bench_dwave100.json 2> $OUT/bench_dwave100.err
ls $OUT $OUT/pmc
