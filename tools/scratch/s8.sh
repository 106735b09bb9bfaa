#!/bin/bash
set -o pipefail
O=gpurun_out/s8; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
tools/profile.sh r05_rfft --workload rfft > gpurun_out/prof_r05_rfft.log 2>&1 || { tail -20 gpurun_out/prof_r05_rfft.log; exit 1; }
tools/profile.sh r05_pconv --workload pconv > gpurun_out/prof_r05_pconv.log 2>&1 || { tail -20 gpurun_out/prof_r05_pconv.log; exit 1; }
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python tools/size_sweep.py 2>&1 | grep -v amdgpu.ids > $O/sizes.txt; cat $O/sizes.txt
python tools/time_any.py 2>/dev/null | grep -v amdgpu.ids > $O/any.txt; cat $O/any.txt
python __graft_entry__.py smoke 2>&1 | tail -2
