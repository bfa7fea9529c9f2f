#!/bin/bash
# where the complex K7b-OS spends its time: timing line + a few SQ counters (one rocprofv3 pass per group)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4cos; rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python scratch/kbench.py "os4=" --model texture --kind z4 --vectors 4 --steps 63 > $OUT/time_texture.log 2>&1
python scratch/kbench.py "dict2=" --model peierls --kind z4 --vectors 2 --steps 63 > $OUT/time_peierls.log 2>&1
python scratch/kbench.py "dict4=BODGE_AMD_SWEEP_LANES=4" --model peierls --kind z4 --vectors 4 --steps 63 >> $OUT/time_peierls.log 2>&1
grep -h "^os4\|^dict" $OUT/time_texture.log $OUT/time_peierls.log
cd /tmp && export TMPDIR=/tmp
i=0
for group in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  for m in texture peierls; do
    vec=4; [ $m = peierls ] && vec=2
    timeout -k 10 200 rocprofv3 --pmc $group --kernel-trace --output-format csv -d $OUT/g${i}_$m -- python3 $GRAFT_REPO_ROOT/scratch/kbench.py "x=" --model $m --kind z4 --vectors $vec --steps 63 --rounds 1 > $OUT/g${i}_$m.log 2>&1 || { tail -5 $OUT/g${i}_$m.log; echo "group '$group' failed"; }
  done
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections, os
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r4cos"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/g*/*/*counter_collection.csv"):
    m = f.split("/g")[1].split("/")[0].split("_")[1]
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "cheb_sweep3" not in name: continue
        key = m + " " + name.split("(")[0].replace("void bdg::", "")
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fh:
    for k, d in sorted(acc.items()):
        print(k, file=fh); print(k)
        for c, v in sorted(d.items()):
            line = f"   {c:28s} mean per launch {sum(v)/len(v):.5g}  ({len(v)} launches)"
            print(line, file=fh); print(line)
PY
find $OUT -name "*.csv" -delete
