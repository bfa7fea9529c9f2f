#!/bin/bash
# tests + bench + rocprofv3 stats + PMC passes; outputs under gpurun_out/full/
set -o pipefail
OUT=$PWD/gpurun_out/full; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q --durations=5 > $OUT/pytest_gpu.log 2>&1 || { tail -20 $OUT/pytest_gpu.log; exit 1; }
timeout -k 10 400 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail $OUT/bench.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 64 --warmup 4 --cpu-seconds 0 > $OUT/stats.log 2>&1 || { tail $OUT/stats.log; exit 1; }
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -- python3 $R/bench.py --steps 8 --warmup 2 --cpu-seconds 0 > $OUT/pmc_$c.log 2>&1 || { tail $OUT/pmc_$c.log; exit 1; }
done
cd $R
find $OUT -name "*_agent_info.csv" -delete
find $OUT -size +8M -delete
tail -3 $OUT/pytest_gpu.log; cat $OUT/bench.json
