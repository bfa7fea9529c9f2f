#!/bin/bash
# bench line, rocprofv3 kernel stats of the same command, PMC traffic passes, N=2 rehearsals on one GPU
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c; rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python3 bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err || { echo bench failed; tail -5 $OUT/bench_n1.err; exit 1; }
python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 > $OUT/bench_n1_steps20.json 2>> $OUT/bench_n1.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-seconds 0 > $OUT/bench_under_rocprof.json 2> $OUT/stats.err || echo "stats run failed"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --cpu-seconds 0 > $OUT/pmc_$c.json 2> $OUT/pmc_$c.err || echo "pmc $c failed"
done
cd $GRAFT_REPO_ROOT
python3 tools/pmc_traffic.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE --workload "1000x1000x1 R=8" --out $OUT/traffic.json > /dev/null
# N=2 on one GPU: RCCL must refuse the duplicate device -> exit code 3 without --allow-gloo, host fallback with it
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 16 --warmup 2 --lattice 400,400,1 > $OUT/n2_strict.out 2> $OUT/n2_strict.err; echo "n2 strict rc=$?" > $OUT/n2.rc
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29631 bench.py --gpus 2 --steps 16 --warmup 2 --lattice 400,400,1 --allow-gloo > $OUT/n2_allow.out 2> $OUT/n2_allow.err; echo "n2 allow rc=$?" >> $OUT/n2.rc
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
find $OUT -name "*_results.db" -delete; find $OUT -name "*agent_info.csv" -delete
du -sh $OUT
