#!/bin/bash
A=tools/ab; O=gpurun_out/s10; mkdir -p $O
python -m pytest tests/test_gpu_fft.py tests/test_gpu_misc.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for w in rfft16384 rfft32768 rfft8192 c2c32768 c2c8192 rfft65536 c2c4096 rfft4096; do python tools/ab_multi.py $w prerow=$A/libclfft_prerow.so 2>/dev/null | grep -v amdgpu.ids; done > $O/ab_final.txt; cat $O/ab_final.txt
