#!/bin/bash
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 tools/handoff_test.hip -o /tmp/handoff || exit 1
for M in 0 1 2; do
  /tmp/handoff $M | tail -1
  for C in FETCH_SIZE WRITE_SIZE; do
    D=/tmp/ho_${M}_$C
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -- /tmp/handoff $M > /dev/null 2>&1
    python3 - <<PY
import csv,glob
v=[float(r["Counter_Value"]) for f in glob.glob("$D/**/*counter_collection.csv",recursive=True) for r in csv.DictReader(open(f)) if "k_pair" in r["Kernel_Name"]]
print("  mode $M $C mean %.0f KB (2 GiB alg = 2097152 KB; FETCH_SIZE reads half)"%(sum(v)/len(v)))
PY
  done
done
