#!/bin/bash
set -o pipefail
O=gpurun_out/s12; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python tools/size_sweep.py 2>&1 | grep -v amdgpu.ids > $O/sizes.txt; cat $O/sizes.txt
python __graft_entry__.py smoke 2>&1 | tail -1
python - <<'PY'
import json
j=json.load(open("gpurun_out/s12/bench.json"))
print("c2c", j["ms_per_step"], j["roofline"]["frac"], j["membench"])
for k,v in j["config"]["other_workloads"].items(): print(k, round(v["ms_per_step"],5), round(v["roofline"]["frac"],4))
PY
