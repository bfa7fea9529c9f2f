#!/bin/bash
# SQ / TCC / fabric counters of the recurrence kernels the bench launches (sweep + one-step forms),
# one rocprofv3 pass per group; summary -> gpurun_out/pmc_sweep/summary.txt
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_sweep; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for group in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $group --kernel-trace --output-format csv -d $OUT/g$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --cpu-seconds 0 "$@" > $OUT/g$i.log 2>&1 || { tail -3 $OUT/g$i.log; echo "group '$group' failed"; }
done
cd $GRAFT_REPO_ROOT
find $OUT -name "*_agent_info.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
python3 - > $OUT/summary.txt <<'PY'
import csv, glob, collections, os, re
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_sweep"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/g*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "cheb_s" not in name and "cheb_roll" not in name: continue
        key = re.sub(r"(, \d)?, (true|false)>$", ">", name.split("(")[0].replace("void bdg::", "")) if "cheb_sweep" in name else name.split("(")[0].replace("void bdg::", "")
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:30s} mean per launch {sum(v)/len(v):.4g}  ({len(v)} launches)")
PY
cat $OUT/summary.txt
