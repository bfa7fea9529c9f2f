#!/bin/bash
# kernel time breakdown of the two-stage eigenpair route (rocprofv3 --kernel-trace --stats), L x L lattice
L=${1:-50}
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_twostage_prof; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/scratch/r4_twostage_vectors.py $L > $OUT/run.log 2>&1
tail -2 $OUT/run.log
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
cp $f $GRAFT_REPO_ROOT/gpurun_out/r04_kernel_stats_twostage_vectors_L$L.csv
rm -rf $OUT
