#!/bin/bash
A=tools/ab; O=gpurun_out/s9; mkdir -p $O
for w in rfft16384 c2c8192 rfft32768 rfft65536 c2c16384 rfft8192 c2c4096 c2c1024 rfft2048; do python tools/ab_multi.py $w ldssingle=$A/libclfft_ldssingle.so 2>/dev/null | grep -v amdgpu.ids; done > $O/ab_ldssingle.txt; cat $O/ab_ldssingle.txt
python tools/ab_multi.py c2c32768 s4single=$A/libclfft_s4single.so 2>/dev/null | grep -v amdgpu.ids > $O/ab_4step.txt; cat $O/ab_4step.txt
