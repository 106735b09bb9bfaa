#!/bin/bash
set -o pipefail
O=gpurun_out/s1; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
A=tools/ab
python tools/ab_multi.py c2c base=$A/libclfft_base.so nods=$A/libclfft_nods.so ds1=$A/libclfft_ds1.so earlyns=$A/libclfft_earlyns.so early=$A/libclfft_early.so tw=$A/libclfft_tw.so fs=$A/libclfft_fs.so > $O/ab_c2c.txt 2>&1 && cat $O/ab_c2c.txt
python tools/ab_multi.py rfft base=$A/libclfft_base.so row16=$A/libclfft_row16.so > $O/ab_rfft.txt 2>&1 && cat $O/ab_rfft.txt
python tools/ab_multi.py rfft131072 base=$A/libclfft_base.so nods=$A/libclfft_nods.so > $O/ab_rfft131072.txt 2>&1 && cat $O/ab_rfft131072.txt
for n in 32768 65536; do python tools/ab_multi.py rfft$n base=$A/libclfft_base.so > $O/ab_rfft$n.txt 2>&1 && cat $O/ab_rfft$n.txt; done
python tools/ab_multi.py c2c8192 base=$A/libclfft_base.so > $O/ab_c2c8192.txt 2>&1 && cat $O/ab_c2c8192.txt
python tools/ab_multi.py c2c16384 base=$A/libclfft_base.so > $O/ab_c2c16384.txt 2>&1 && cat $O/ab_c2c16384.txt
tools/lds_pmc2.sh tree > /dev/null 2>&1; cat gpurun_out/ldspmc_tree.txt
tools/lds_pmc2.sh base tools/ab/libclfft_base.so > /dev/null 2>&1; cat gpurun_out/ldspmc_base.txt
