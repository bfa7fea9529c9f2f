#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3td; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/scratch/tridiag_vectors_check.py swave50_zeeman > $OUT/run.log 2> $OUT/err.log
cat $OUT/run.log
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv; rm -rf $OUT/stats
cut -d, -f1-5 $OUT/kernel_stats.csv | cut -c1-160 | head -12
