#!/bin/bash
set -o pipefail
O=gpurun_out/s4; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
python tools/host_path2.py > $O/host_path2.txt 2>&1; cat $O/host_path2.txt
A=tools/ab
for w in c2c256 c2c512 c2c1024 c2c2048 c2c4096 rfft512 rfft1024 rfft2048 rfft4096 rfft8192 rfft16384; do python tools/ab_multi.py $w pretab=$A/libclfft_pretab.so 2>/dev/null | grep -v amdgpu.ids; done > $O/ab_tabs.txt; cat $O/ab_tabs.txt
( echo "== release on the arrival add (library)"; python tools/pconv_latency.py; python tools/dconv_latency.py; export CLFA_LIB_PATH=$PWD/tools/ab/libclfft_norel.so; echo "== relaxed add (CLFA_HANDOVER_RELEASE=0)"; python tools/pconv_latency.py; python tools/dconv_latency.py ) 2>&1 | grep -v amdgpu.ids > $O/handover.txt; cat $O/handover.txt
python tools/batch_sweep.py 2>&1 | grep -v amdgpu.ids > $O/batch_sweep.txt; cat $O/batch_sweep.txt
tools/profile.sh r05_rfft131072 --workload rfft131072 > gpurun_out/prof_r05_rfft131072.log 2>&1 || tail -5 gpurun_out/prof_r05_rfft131072.log
tail -3 gpurun_out/prof_r05_rfft131072.log | cut -c1-300
python tools/size_sweep.py 2>&1 | grep -v amdgpu.ids > $O/sizes.txt; cat $O/sizes.txt
