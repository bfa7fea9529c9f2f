#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_ea; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for group in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $group --kernel-trace --output-format csv -d $OUT/g$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --cpu-seconds 0 > $OUT/g$i.log 2>&1 || { grep -i "error\|invalid\|not" $OUT/g$i.log | head -3; echo "group '$group' failed"; }
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections, os
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_ea"
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
