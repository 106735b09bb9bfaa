#!/bin/bash
# usage: tools/lds_pmc3.sh -> gpurun_out/ldspmc3.txt : LDS / issue counters per kernel for the other sizes
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/ldspmc3.txt; : > $OUT
for C in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  D=/tmp/lp3_$(echo $C | tr ' ' '_' | cut -c1-30); rm -rf $D
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -- python3 tools/lds_pmc3.py > $D.log 2>&1 || { echo "pmc pass failed: $C"; tail -3 $D.log; }
  python3 - >> $OUT <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$D/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "clfa" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:58]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    d={c: sum(v)/len(v) for c, v in acc[k].items()}
    extra=""
    if "SQ_LDS_IDX_ACTIVE" in d and d["SQ_LDS_IDX_ACTIVE"]>0: extra=" conflict ratio %.3f"%(d["SQ_LDS_BANK_CONFLICT"]/d["SQ_LDS_IDX_ACTIVE"])
    if "SQ_WAVE_CYCLES" in d: extra=" wait_any %.2f wait_inst %.2f active %.2f"%(d["SQ_WAIT_ANY"]/d["SQ_WAVE_CYCLES"],d["SQ_WAIT_INST_ANY"]/d["SQ_WAVE_CYCLES"],d["SQ_ACTIVE_INST_ANY"]/d["SQ_WAVE_CYCLES"])
    print(k, {c: "%.4g" % v for c, v in d.items()}, extra)
PY
done
cat $OUT
