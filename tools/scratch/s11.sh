#!/bin/bash
A=tools/ab; O=gpurun_out/s11; mkdir -p $O
python -m pytest tests/test_gpu_fft.py tests/test_gpu_misc.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for w in c2c4096 rfft8192 c2c8192 rfft4096; do python tools/ab_multi.py $w prelane12=$A/libclfft_prelane12.so 2>/dev/null | grep -v amdgpu.ids; done > $O/ab_lane12.txt; cat $O/ab_lane12.txt
