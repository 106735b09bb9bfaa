#!/bin/bash
# usage: tools/lds_pmc2.sh <tag> [lib.so]   -> gpurun_out/ldspmc_<tag>.txt : LDS / issue counters per kernel (means per launch)
export TMPDIR=/tmp
TAG=$1
[ -n "$2" ] && export CLFA_LIB_PATH=$PWD/$2
OUT=$PWD/gpurun_out/ldspmc_$TAG.txt; : > $OUT
for C in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU"; do
  D=/tmp/lp2_$TAG_$(echo $C | tr ' ' '_' | cut -c1-30); rm -rf $D
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -- python3 tools/lds_pmc2.py > $D.log 2>&1 || { echo "pmc pass failed: $C"; tail -3 $D.log; }
  python3 - >> $OUT <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$D/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "clfa" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k, {c: "%.4g" % (sum(v)/len(v)) for c, v in acc[k].items()})
PY
done
cat $OUT
