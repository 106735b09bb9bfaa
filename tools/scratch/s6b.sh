#!/bin/bash
O=gpurun_out/s6; mkdir -p $O
for k in 1 2 3; do python -m pytest tests -m gpu -q > $O/pytest_$k.log 2>&1; tail -2 $O/pytest_$k.log; grep -E "AssertionError: pinned|FAILED" $O/pytest_$k.log | head -5; done
