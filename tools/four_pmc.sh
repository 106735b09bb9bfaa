#!/bin/bash
export TMPDIR=/tmp
for C in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY"; do
  D=/tmp/fp_$(echo $C | tr ' ' '_' | cut -c1-30); rm -rf $D
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("$D/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_fft_4step<16, true" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print({c: "%.3g" % (sum(v)/len(v)) for c, v in acc.items()})
PY
done
