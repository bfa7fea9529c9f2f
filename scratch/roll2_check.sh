#!/bin/bash
# validation + evidence for the 2-lane rolling kernel: parity tests, config-4 tests, bench line, rocprofv3 stats, PMC traffic
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/roll2; rm -rf $OUT; mkdir -p $OUT
cd $R
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -m gpu -x -k "multi_step_sweep or config4 or random_periodic or step_aside" 2>&1 | tail -3 | tee $OUT/pytest.log
python3 bench.py --model dwave --lattice 100,100,100 --cpu-seconds 0 > $OUT/bench_dwave100.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_dwave -- python3 $R/bench.py --model dwave --lattice 100,100,100 --cpu-seconds 0 > $OUT/bench_dwave100_under_rocprof.json 2> $OUT/stats_dwave.err || echo "stats dwave run failed"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_dwave_$c -- python3 $R/bench.py --steps 32 --warmup 2 --cpu-seconds 0 --model dwave --lattice 100,100,100 > $OUT/pmc_dwave_$c.json 2> $OUT/pmc_dwave_$c.err || echo "pmc dwave $c failed"
done
cd $R
cp profiles/traffic.json $OUT/traffic.json
python3 tools/pmc_traffic.py $OUT/pmc_dwave_FETCH_SIZE $OUT/pmc_dwave_WRITE_SIZE --workload "100x100x100 R=8" --out $OUT/traffic.json > /dev/null
cp $(find $OUT/stats_dwave -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_bench_dwave100.csv
mkdir -p $OUT/pmc; for d in pmc_dwave_FETCH_SIZE pmc_dwave_WRITE_SIZE; do cp $(find $OUT/$d -name "*counter_collection.csv" | head -1) $OUT/pmc/${d#pmc_}_counter_collection.csv; done
rm -rf $OUT/stats_dwave $OUT/pmc_dwave_FETCH_SIZE $OUT/pmc_dwave_WRITE_SIZE
python3 -c "
import json; d=json.load(open('$OUT/bench_dwave100.json')); r=d['roofline']; print('dwave100:', round(d['value']), r['kernel'], 'launch_ms', r['launch_ms'], 'frac', round(r['frac'],3), 'traffic', r['traffic'], 'one-step', d.get('one_step_kernels',{}).get('value'))
t=json.load(open('$OUT/traffic.json')); print({k: round(v['traffic_bytes_per_launch']/1e9,4) for k,v in t.items() if 'roll3' in k})"
