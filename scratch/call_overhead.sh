#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/overhead; rm -rf $OUT; mkdir -p $OUT
python3 $GRAFT_REPO_ROOT/scratch/call_overhead.py | tee $OUT/wall.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --hip-trace --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/scratch/call_overhead.py > $OUT/trace.log 2>&1
find $OUT/trace -name "*kernel_trace.csv" -exec cp {} $OUT/kernel_trace.csv \;
find $OUT/trace -name "*hip_api_trace.csv" -exec cp {} $OUT/hip_trace.csv \;
rm -rf $OUT/trace
ls -la $OUT
