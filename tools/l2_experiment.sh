#!/bin/bash
# does the four-step scratch stay in the XCD L2 when few workgroups are in flight?
export TMPDIR=/tmp
mkdir -p $PWD/gpurun_out/l2exp
for G in 16 32 48 64 256; do
  for C in FETCH_SIZE WRITE_SIZE; do
    D=$PWD/gpurun_out/l2exp/g${G}_$C
    CLFA_4STEP_GRID=$G rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --batch 1024 --variant 1 > $D.log 2>&1
  done
  python3 - <<PY
import csv,glob
for c in ("FETCH_SIZE","WRITE_SIZE"):
    vals=[]
    for f in glob.glob("gpurun_out/l2exp/g${G}_%s/**/*counter_collection.csv"%c, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_fft_4step" in r["Kernel_Name"]: vals.append(float(r["Counter_Value"]))
    vals=sorted(vals)[len(vals)//2:]   # drop the small parity-check launch
    print("grid ${G} %s median-of-upper KB = %.0f  (1 GiB alg each way = 1048576 KB)"%(c, sum(vals)/max(1,len(vals))))
PY
  CLFA_4STEP_GRID=$G python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --batch 1024 --variant 1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('grid $G  ms/launch %.3f  alg GB/s %.0f'%(r['roofline']['avg_launch_ms'], r['roofline']['achieved']))"
done
