#!/bin/bash
# dump the gfx950 ISA of one kernel: tools/isa.sh <mangled-name-regex> [file.hip]
cd "$(dirname "$0")/../opencl_fft_amd/csrc" || exit 1
F=${2:-fft_kernels.hip}
/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC -fno-slp-vectorize --offload-arch=gfx950 -save-temps=obj -c $F -o /tmp/isa_tmp.o 2>/dev/null || exit 1
S=/tmp/$(basename $F .hip)-hip-amdgcn-amd-amdhsa-gfx950.s
awk "/^$1:/,/s_endpgm/" $S
