#!/bin/bash
set -o pipefail
tools/profile.sh r05_c2c > gpurun_out/prof_r05_c2c.log 2>&1 || { tail -20 gpurun_out/prof_r05_c2c.log; exit 1; }
tail -5 gpurun_out/prof_r05_c2c.log
tools/profile.sh r05_rfft --workload rfft > gpurun_out/prof_r05_rfft.log 2>&1 || { tail -20 gpurun_out/prof_r05_rfft.log; exit 1; }
tail -5 gpurun_out/prof_r05_rfft.log
