#!/bin/bash
set -o pipefail
O=gpurun_out/s5; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
A=tools/ab
for w in rfft16384 c2c8192 rfft32768 rfft65536 c2c16384 c2c32768 rfft8192; do python tools/ab_multi.py $w nosigma=$A/libclfft_nosigma.so 2>/dev/null | grep -v amdgpu.ids; done > $O/ab_sigma.txt; cat $O/ab_sigma.txt
python tools/host_path2.py pr1=$A/libclfft_pr1.so pr2=$A/libclfft_pr2.so zc512=$A/libclfft_zc512.so 2>&1 | grep -v amdgpu.ids > $O/host_path2.txt; cat $O/host_path2.txt
tools/lds_pmc2.sh tree > /dev/null 2>&1; cat gpurun_out/ldspmc_tree.txt
