#!/bin/bash
set -o pipefail
O=gpurun_out/s3; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
python -m pytest tests/test_gpu_conv_accuracy.py -m gpu -q -s 2>&1 | grep ACCURACY > $O/accuracy.txt; cat $O/accuracy.txt
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python - <<'PY'
import json
j=json.load(open("gpurun_out/s3/bench.json"))
print("c2c", j["ms_per_step"], j["roofline"]["frac"], j["reference_opencl_same_gpu"])
for k,v in j["config"]["other_workloads"].items(): print(k, v["ms_per_step"], v["roofline"]["frac"], v.get("reference_opencl_same_gpu"))
PY
python tools/host_breakdown.py tools/ab/libclfft_zc512.so tools/ab/libclfft_zc1024.so > $O/host_path.txt 2>&1; cat $O/host_path.txt
( echo "== release on the arrival add (library)"; python tools/pconv_latency.py; python tools/dconv_latency.py; export CLFA_LIB_PATH=$PWD/tools/ab/libclfft_norel.so; echo "== relaxed add (CLFA_HANDOVER_RELEASE=0)"; python tools/pconv_latency.py; python tools/dconv_latency.py ) > $O/handover.txt 2>&1; cat $O/handover.txt
python tools/rt_sweep.py 3 60 > $O/rt_sweep.txt 2>&1; cat $O/rt_sweep.txt
