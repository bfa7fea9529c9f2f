#!/bin/bash
# a few SQ / TCC counters for the headline kernel, one rocprofv3 pass per group
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_sq; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for group in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $group --kernel-trace --output-format csv -d $OUT/g$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --cpu-seconds 0 > $OUT/g$i.log 2>&1 || { tail -5 $OUT/g$i.log; echo "group '$group' failed"; }
done
cd $GRAFT_REPO_ROOT
find $OUT -name "*_agent_info.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete
python3 - <<'PY'
import csv, glob, collections, os
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_sq"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/g*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "cheb_step" not in name: continue
        key = name.split("(")[0].replace("void bdg::", "")
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} mean per launch {sum(v)/len(v):.4g}  ({len(v)} launches)")
PY
