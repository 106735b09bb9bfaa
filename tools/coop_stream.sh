#!/bin/bash
# streaming mode x window of the cooperative kernel: time + L2<->fabric traffic
export TMPDIR=/tmp
for SM in 1 2; do
 for cfg in "512 2 4" "512 3 5" "768 3 6"; do
  set -- $cfg
  export CLFA_COOP_STREAM=$SM CLFA_COOP_GRID=$1 CLFA_COOP_LAG=$2 CLFA_COOP_SLOTS=$3
  T=$(timeout -k 5 100 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --variant 7 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('%.3f ms'%(r['roofline']['avg_launch_ms']))")
  D=/tmp/cs_${SM}_$1_$2; rm -rf $D
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $D -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --variant 7 > /dev/null 2>&1
  F=$(python3 - <<PY
import csv,glob
v=sorted(float(r["Counter_Value"]) for f in glob.glob("$D/**/*counter_collection.csv",recursive=True) for r in csv.DictReader(open(f)) if "k_fft_coop" in r["Kernel_Name"])
v=v[len(v)//4:]
print("read %.2f GiB"%(2*1024*sum(v)/len(v)/2**30))
PY
)
  echo "stream $SM grid $1 lag $2 slots $3 : $T  $F (input alone = 2.00 GiB)"
 done
done
