#!/bin/bash
# usage: tools/pmc_sizes.sh <lo> <hi> : FETCH_SIZE / WRITE_SIZE per kernel of tools/size_sweep.py <lo> <hi>
# (separate --pmc passes; FETCH_SIZE doubled for 128-byte requests as MI355X_MICROARCH.md prescribes for gfx950)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_sizes; mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$C -- python3 tools/size_sweep.py $1 $2 > $OUT/$C.log 2>&1
done
python3 - <<PY
import csv,glob,collections,re
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        if "clfa::" not in n: continue
        n=re.sub(r"\(.*","",n.replace("void clfa::",""))
        acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("kernel | launches | read GiB (FETCH_SIZE x 2) | write GiB | total GiB per launch (algorithmic: 2.00)")
for n,c in sorted(acc.items()):
    f=c.get("FETCH_SIZE"); w=c.get("WRITE_SIZE")
    if not f or not w: continue
    fb=2*1024*sum(f)/len(f); wb=1024*sum(w)/len(w)
    print("%-44s %4d  %.3f  %.3f  %.3f" % (n[:44], len(f), fb/2**30, wb/2**30, (fb+wb)/2**30))
PY
