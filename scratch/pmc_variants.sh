#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, one counter per pass) and timing of K7b under a few env variants
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcv; rm -rf $OUT; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
python3 $R/scratch/sweep_ab.py "stream1:" "stream0:BODGE_AMD_SWEEP_STREAM=0" "stream3:BODGE_AMD_SWEEP_STREAM=3" "stream2:BODGE_AMD_SWEEP_STREAM=2" "stream5:BODGE_AMD_SWEEP_STREAM=5" 2>&1 | tee $OUT/timing.log
cd /tmp && export TMPDIR=/tmp
for v in "stream1:" "stream0:BODGE_AMD_SWEEP_STREAM=0" "stream3:BODGE_AMD_SWEEP_STREAM=3" "stream2:BODGE_AMD_SWEEP_STREAM=2"; do
  name=${v%%:*}
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/${name}_$c -- python3 $R/scratch/sweep_ab.py "$v" > $OUT/${name}_$c.log 2>&1 || echo "$name $c failed"
  done
  echo "== $name" | tee -a $OUT/summary.txt
  python3 $R/tools/pmc_traffic.py $OUT/${name}_FETCH_SIZE $OUT/${name}_WRITE_SIZE --workload "$name" --out $OUT/traffic_$name.json | tee -a $OUT/summary.txt
  rm -rf $OUT/${name}_FETCH_SIZE $OUT/${name}_WRITE_SIZE
done
