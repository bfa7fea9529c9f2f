#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4buf
{
echo "# scratch/r4_bench_ctx.py: the same 8-vector call on a matrix with position-dependent on-site terms, as the only handle of"
echo "# the process and as its second handle (as in bench.py's comparison passes)"
python scratch/r4_bench_ctx.py other_first 2>&1 | grep -v Warning
echo ---- main solver first
python scratch/r4_bench_ctx.py main_first 2>&1 | grep -v Warning
} > gpurun_out/r4buf/ctx.log
cat gpurun_out/r4buf/ctx.log
bash scratch/r4_share3.sh
