#!/bin/bash
# sweep grid x lag x slots of the cooperative kernel (timing via bench.py roofline block)
for G in 512 768 1024; do
 for L in 1 2 3 4; do
  for S in 4 6 8 10; do
    R=$(CLFA_COOP_GRID=$G CLFA_COOP_LAG=$L CLFA_COOP_SLOTS=$S timeout -k 5 120 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --variant 7 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('%.3f ms  alg %.0f GB/s  parity %.1e'%(r['roofline']['avg_launch_ms'], r['roofline']['achieved'], r['config']['parity_relL2_vs_oracle']))")
    echo "grid $G lag $L slots $S : $R"
  done
 done
done
