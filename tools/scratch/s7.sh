#!/bin/bash
set -o pipefail
O=gpurun_out/s7; mkdir -p $O
timeout -k 10 300 python tools/stress_pinned.py 3000 2>&1 | grep -v amdgpu.ids | tail -5 || exit 1
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
python tools/host_path2.py 2>&1 | grep -v amdgpu.ids > $O/host_path_final.txt; cat $O/host_path_final.txt
