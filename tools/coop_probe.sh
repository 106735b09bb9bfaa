for cfg in "256 1 3" "256 1 4" "384 1 4" "384 2 5" "512 1 4" "512 2 5" "512 3 5"; do
  set -- $cfg
  R=$(CLFA_COOP_GRID=$1 CLFA_COOP_LAG=$2 CLFA_COOP_SLOTS=$3 timeout -k 5 100 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --variant 7 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('%.3f ms  alg %.0f GB/s'%(r['roofline']['avg_launch_ms'], r['roofline']['achieved']))")
  echo "grid $1 lag $2 slots $3 : $R"
done
