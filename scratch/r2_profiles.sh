#!/bin/bash
# Round-2 evidence in one call: bench lines (configs[2] and configs[3] lattices), rocprofv3 kernel
# stats of the same commands, PMC traffic passes (one counter per pass), N = 2 rehearsals on one GPU.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2p; rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python3 bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err || { echo bench failed; tail -5 $OUT/bench_n1.err; exit 1; }
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_n1_steps20_warmup5.json 2>> $OUT/bench_n1.err
python3 bench.py --model dwave --lattice 100,100,100 --cpu-seconds 0 > $OUT/bench_dwave100.json 2>> $OUT/bench_n1.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-seconds 0 > $OUT/bench_n1_under_rocprof.json 2> $OUT/stats.err || echo "stats run failed"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_dwave -- python3 $GRAFT_REPO_ROOT/bench.py --model dwave --lattice 100,100,100 --cpu-seconds 0 > $OUT/bench_dwave100_under_rocprof.json 2> $OUT/stats_dwave.err || echo "stats dwave run failed"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 63 --warmup 3 --cpu-seconds 0 > $OUT/pmc_$c.json 2> $OUT/pmc_$c.err || echo "pmc $c failed"
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_dwave_$c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 32 --warmup 2 --cpu-seconds 0 --model dwave --lattice 100,100,100 > $OUT/pmc_dwave_$c.json 2> $OUT/pmc_dwave_$c.err || echo "pmc dwave $c failed"
done
cd $GRAFT_REPO_ROOT
cp profiles/traffic.json $OUT/traffic.json
python3 tools/pmc_traffic.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE --workload "1000x1000x1 R=8" --out $OUT/traffic.json > /dev/null
python3 tools/pmc_traffic.py $OUT/pmc_dwave_FETCH_SIZE $OUT/pmc_dwave_WRITE_SIZE --workload "100x100x100 R=8" --out $OUT/traffic.json > /dev/null
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 16 --warmup 2 --lattice 400,400,1 > $OUT/n2_strict.out 2> $OUT/n2_strict.err; echo "N=2 on one GPU, no --allow-gloo: rc=$?" > $OUT/n2.rc
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29631 bench.py --gpus 2 --steps 16 --warmup 2 --lattice 400,400,1 --allow-gloo > $OUT/n2_allow.out 2> $OUT/n2_allow.err; echo "N=2 on one GPU, --allow-gloo: rc=$?" >> $OUT/n2.rc
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_bench_default.csv
cp $(find $OUT/stats_dwave -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_bench_dwave100.csv
mkdir -p $OUT/pmc; for d in pmc_FETCH_SIZE pmc_WRITE_SIZE pmc_dwave_FETCH_SIZE pmc_dwave_WRITE_SIZE; do cp $(find $OUT/$d -name "*counter_collection.csv" | head -1) $OUT/pmc/${d#pmc_}_counter_collection.csv; done
rm -rf $OUT/stats $OUT/stats_dwave $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_dwave_FETCH_SIZE $OUT/pmc_dwave_WRITE_SIZE
du -sh $OUT
