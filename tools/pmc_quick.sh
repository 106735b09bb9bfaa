#!/bin/bash
# usage: tools/pmc_quick.sh <tag> <bench args...> : FETCH_SIZE / WRITE_SIZE / L2 hit for the FFT kernels
export TMPDIR=/tmp
TAG=$1; shift
OUT=$PWD/gpurun_out/pmcq_$TAG; mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  D=$OUT/$(echo $C | tr ' ' '_')
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline "$@" > $D.log 2>&1
done
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_fft" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    v=sorted(v)[len(v)//4:]   # drop the tiny parity-check launches
    print("%-14s mean %.0f  (n=%d)"%(k, sum(v)/len(v), len(v)))
f=acc.get("FETCH_SIZE"); w=acc.get("WRITE_SIZE")
if f and w:
    f=sorted(f)[len(f)//4:]; w=sorted(w)[len(w)//4:]
    fb=2*1024*sum(f)/len(f); wb=1024*sum(w)/len(w)
    print("read %.2f GiB (FETCH_SIZE x2, gfx950 correction)  write %.2f GiB  total %.2f GiB per launch"%(fb/2**30, wb/2**30,(fb+wb)/2**30))
PY
