#!/bin/bash
# round 4: bench records + profiles of the final code.  rocprofv3 kernel stats of the bench at the driver's flags and at 256 steps,
# one PMC pass per counter (FETCH_SIZE / WRITE_SIZE cannot share a pass) with full-length chunks, the two bench lines themselves.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4p; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for flags in "--steps 20 --warmup 5" "--steps 256 --warmup 8"; do
  tag=$(echo $flags | tr -d ' -' )
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$tag -- python3 $GRAFT_REPO_ROOT/bench.py $flags --cpu-seconds 0 --user-calls 0 > $OUT/bench_under_rocprof_$tag.json 2> $OUT/stats_$tag.err || echo "stats $tag failed"
  cp $(find $OUT/stats_$tag -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_bench_$tag.csv
  rm -rf $OUT/stats_$tag
  echo "[records] stats $tag done"
done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 63 --warmup 3 --cpu-seconds 0 --user-calls 0 > $OUT/pmc_$c.json 2> $OUT/pmc_$c.err || echo "pmc $c failed"
  echo "[records] pmc $c done"
done
cd $GRAFT_REPO_ROOT
cp profiles/traffic.json $OUT/traffic.json
python3 tools/pmc_traffic.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE --workload "1000x1000x1 R=8" --out $OUT/traffic.json > /dev/null
mkdir -p $OUT/pmc; for d in pmc_FETCH_SIZE pmc_WRITE_SIZE; do cp $(find $OUT/$d -name "*counter_collection.csv" | head -1) $OUT/pmc/${d#pmc_}_counter_collection.csv; done
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
cp $OUT/traffic.json profiles/traffic.json   # (the bench lines below report the traffic measured on this code)
python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err && echo "[records] bench 256 done" &&
python bench.py --steps 20 --warmup 5 > $OUT/bench_n1_steps20_warmup5.json 2> $OUT/bench_n1_steps20_warmup5.err &&
python - <<'PY'
import json, os
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r4p/"
for f in ("bench_n1", "bench_n1_steps20_warmup5"):
    r = json.loads(open(out + f + ".json").read().strip().splitlines()[-1]); ro = r["roofline"]
    print(f, round(r["value"]), "frac", round(ro["frac"],4), "launch_ms", round(ro["launch_ms"],5), "window", round(ro["window_ms"],4), "grid", ro["grid"], "traffic", ro["traffic"], "cpu", round(r["cpu_baseline"]["value"]), r["cpu_baseline"]["cores"])
    for k in ("two_step_kernels","one_step_kernels","streamed_blocks_kernels","complex128_kernels","streamed_bonds_kernels","complex128_sweep_kernels"):
        if r.get(k): print("   ", k, round(r[k]["value"]), round(r[k]["frac"],3), r[k]["kernel"], r[k].get("streams"))
    print("   user_facing_calls", json.dumps(r.get("user_facing_calls"))[:1500])
PY
grep "cheb_sweep3<bdg::RealPHMode, 2" $OUT/kernel_stats_bench_steps20warmup5.csv | cut -c1-150
