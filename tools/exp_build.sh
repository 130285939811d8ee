#!/bin/bash
# Timing experiments: builds variants of libspicey_hip.so with -DSPICEY_EXP=<bits> into build/exp/ (git-ignored,
# shipped to the GPU box).  Use with SPICEY_HIP_LIB=build/exp/libspicey_hip_expN.so python tools/perf_probe.py ...
# Results of these variants are WRONG by construction (work is skipped); only their timing is of interest.
set -e
cd "$(dirname "$0")/../spicey_amd/csrc"
mkdir -p ../../build/exp
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSPICEY_EXP=$n $SPICEY_EXP_FLAGS -shared -o ../../build/exp/libspicey_hip_exp$n.so spicey_abi.cpp symbolic.cpp kernels.hip &
done
wait
ls -la ../../build/exp
