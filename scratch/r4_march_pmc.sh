#!/bin/bash
# round 4: SQ / TCC / traffic counters of cheb_march3 (BODGE_AMD_MARCH=1 tickets, 3 fixed) next to cheb_sweep3 (0), one rocprofv3 pass per
# counter group and mode; 1000x1000 s-wave+Zeeman, 8 vectors, 63 steps per call.  Summary -> gpurun_out/r04_march_pmc.txt
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_march_pmc; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mode in 0 1 3; do
  export BODGE_AMD_MARCH=$mode
  i=0
  for group in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --pmc $group --kernel-trace --output-format csv -d $OUT/m${mode}_g$i -- python3 $GRAFT_REPO_ROOT/scratch/kbench.py "x=" --vectors 8 --steps 63 --rounds 2 > $OUT/m${mode}_g$i.log 2>&1 || { tail -3 $OUT/m${mode}_g$i.log; echo "mode $mode group '$group' failed"; }
  done
done
cd $GRAFT_REPO_ROOT
python3 - > gpurun_out/r04_march_pmc.txt <<'PY'
import csv, glob, collections, os, re
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r04_march_pmc"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/m*_g*/*/*counter_collection.csv"):
    mode = re.search(r"/m(\d)_g", f).group(1)
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "cheb_sweep3" not in name and "cheb_march3" not in name: continue
        key = "MARCH=" + mode + " " + re.sub(r"\(.*", "", name.replace("void bdg::", ""))
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("counters per launch (cheb_sweep3: one sweep of one lane group = 3 steps x 4 vectors; cheb_march3: 21 sweeps x 2 lane groups)")
for k, d in sorted(acc.items()):
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:30s} mean per launch {sum(v)/len(v):.5g}   sum over the run {sum(v):.5g}  ({len(v)} launches)")
PY
rm -rf $OUT
cat gpurun_out/r04_march_pmc.txt
