#!/bin/bash
# ts_chase kernel time on an L x L lattice for several grids (1 = one wave does every sweep: pure compute time per step)
L=${1:-30}
cd /tmp && export TMPDIR=/tmp
for grid in 1 8 2048; do
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_chase_prof; rm -rf $OUT; mkdir -p $OUT
BODGE_AMD_EIGH_CHASE_GRID=$grid timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/scratch/r4_twostage_check.py $L > $OUT/run.log 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
echo "grid $grid: $(grep ts_chase $f | cut -d, -f1-4)"
rm -rf $OUT
done
