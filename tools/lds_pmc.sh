#!/bin/bash
export TMPDIR=/tmp
for C in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  D=/tmp/lp_$(echo $C | tr ' ' '_' | cut -c1-30); rm -rf $D
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -- python3 tools/lds_pmc.py > /dev/null 2>&1
  python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$D/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_fft_lds" in r["Kernel_Name"]:
            key = "n=4096" if "ILi12" in r["Kernel_Name"] or "<12" in r["Kernel_Name"] else "n=8192"
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k, {c: "%.3g" % (sum(v)/len(v)) for c, v in acc[k].items()})
PY
done
