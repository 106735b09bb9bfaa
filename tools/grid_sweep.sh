#!/bin/bash
V=${1:-10}
for G in 256 320 384 448 512 576 640 768; do
  R=$(CLFA_4STEP_GRID=$G timeout -k 5 100 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --variant $V 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('%.3f ms  alg %.0f GB/s'%(r['roofline']['avg_launch_ms'], r['roofline']['achieved']))")
  echo "variant $V grid $G : $R"
done
