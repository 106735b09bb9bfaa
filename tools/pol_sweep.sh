#!/bin/bash
# dev tool: cache-policy sweep of the resident kernel's global loads / stores (in place and ping-pong out of place).
#   tools/pol_sweep.sh build   (here: one probe binary per policy pair under tools/build/pol/)
#   tools/pol_sweep.sh run     (on the GPU box)
set -e
LD=("" "nt" "sc1" "sc0" "sc0 sc1" "sc1 nt" "sc0 nt" "sc0 sc1 nt")
aux() { a=0; [[ "$1" == *sc0* ]] && a=$((a+1)); [[ "$1" == *nt* ]] && a=$((a+2)); [[ "$1" == *sc1* ]] && a=$((a+16)); echo $a; }
mkdir -p tools/build/pol
if [ "$1" = build ]; then
  n=0
  for l in "${LD[@]}"; do for s in "${LD[@]}"; do
    tag="L$(echo $l | tr ' ' '_')_S$(echo $s | tr ' ' '_')"
    ( /opt/rocm/bin/hipcc -std=c++17 -O3 --offload-arch=gfx950 -fno-slp-vectorize -DPROBE_OOP_ONLY "-DCLFA_LDNT=\" $l\"" "-DCLFA_STNT=\" $s\"" -DCLFA_ST_AUX=$(aux "$s") \
        -I opencl_fft_amd/csrc tools/res16_probe.hip -o tools/build/pol/$tag 2>/dev/null || echo "build failed $tag" ) &
    n=$((n+1)); if [ $((n % 8)) = 0 ]; then wait; fi
  done; done; wait
else
  for round in 1 2; do
  for l in "${LD[@]}"; do for s in "${LD[@]}"; do
    tag="L$(echo $l | tr ' ' '_')_S$(echo $s | tr ' ' '_')"
    [ -x tools/build/pol/$tag ] || continue
    echo "== loads [$l] stores [$s]"
    PROBE_OOP=1 PROBE_PP=1 timeout -k 5 60 tools/build/pol/$tag | grep -E "in place|ping-pong" | head -4
  done; done; done
fi
