#!/bin/bash
# round 4, second part of the records: config 4 (100^3 d-wave) bench lines, the dense two-stage route (timings, residuals, kernel
# statistics of eigenvalues / eigenpairs at n = 10^4, eigenvalues at n = 40000), rate of the 2-lane complex streamed form
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4x; rm -rf $OUT; mkdir -p $OUT
python bench.py --lattice 100,100,100 --model dwave --cpu-seconds 0 --user-calls 0 > $OUT/bench_dwave100.json 2> $OUT/bench_dwave100.err
python bench.py --lattice 100,100,100 --model dwave --vectors-per-gpu 16 --cpu-seconds 0 --user-calls 0 > $OUT/bench_dwave100_16vectors.json 2> $OUT/bench_dwave100_16vectors.err
python - <<'PY'
import json
for f in ("bench_dwave100", "bench_dwave100_16vectors"):
    r = json.loads(open("gpurun_out/r4x/" + f + ".json").read().strip().splitlines()[-1]); ro = r["roofline"]
    print(f, round(r["value"]), "frac", round(ro["frac"], 4), ro["kernel"], "launch_ms", round(ro["launch_ms"], 5), "streams", ro["streams"])
PY
{
for L in 30 50 60; do python scratch/r4_twostage_check.py $L 2>&1 | grep -v Warning | tail -6; done
python scratch/r4_twostage_check.py 100 2>&1 | grep -v Warning | tail -4
for L in 30 50 60; do python scratch/r4_twostage_vectors.py $L 2>&1 | grep -v Warning | tail -6; done
} > $OUT/twostage.log 2>&1
cat $OUT/twostage.log
bash scratch/r4_twostage_prof.sh 50 > $OUT/prof_values.log 2>&1; cp gpurun_out/r04_kernel_stats_twostage_L50.csv $OUT/
bash scratch/r4_twostage_prof_vectors.sh 50 > $OUT/prof_vectors.log 2>&1; cp gpurun_out/r04_kernel_stats_twostage_vectors_L50.csv $OUT/ 2>/dev/null
python scratch/kbench.py "texture_os4=" "texture_os2=BODGE_AMD_SWEEP_LANES=2" --model texture --kind z4 --vectors 8 --steps 63 --rounds 4 2>&1 | grep "^texture_" | cut -c1-190 > $OUT/texture_lanes.log
cat $OUT/texture_lanes.log
