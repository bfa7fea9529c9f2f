#!/bin/bash
# long randomised runs of the final code: stencil kernels (all forms, incl. the streamed ones with 2 / 4 lanes and the complex site
# records) against the one-step kernels; the dense routes against numpy
cd $GRAFT_REPO_ROOT
LOG=gpurun_out/r04_fuzz.log
{
echo "== fuzz_sweep seed 411, 3000 cases"; FUZZ_SEED=411 FUZZ_CASES=3000 timeout -k 10 1000 python tests/fuzz_sweep.py 2>&1 | grep -v "^ok" | tail -8
echo "== fuzz_sweep seed 412, 500 large cases"; FUZZ_SEED=412 FUZZ_CASES=500 FUZZ_SIZE=large timeout -k 10 1000 python tests/fuzz_sweep.py 2>&1 | grep -v "^ok" | tail -8
echo "== fuzz_dense seed 403, 120 cases, default routes"; FUZZ_SEED=403 FUZZ_CASES=120 timeout -k 10 900 python tests/fuzz_dense.py 2>&1 | grep -v "^ok" | tail -6
echo "== fuzz_dense seed 404, 120 cases, two-stage forced"; FUZZ_SEED=404 FUZZ_CASES=120 FUZZ_STAGES=2 timeout -k 10 900 python tests/fuzz_dense.py 2>&1 | grep -v "^ok" | tail -6
} > $LOG 2>&1
cat $LOG
