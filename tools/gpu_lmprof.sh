#!/bin/bash
# development build with per-phase cycle counters of the Lloyd-Max design (not shipped: built into /tmp)
export TMPDIR=/tmp
set -e
cp -r . /tmp/prof_build && cd /tmp/prof_build
touch qvz_amd/csrc/qvz_codebook.hip
make -s HIPFLAGS="--offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Iinclude -Iqvz_amd/csrc -DQVZ_PROFILE" > /dev/null 2>&1
QVZ_CB_DEBUG=1 timeout -k 10 200 python3 bench.py --config 2 --steps 1 --warmup 0 --no-cpu-baseline --no-extras 2>&1 | grep -E "lm_design|generate_codebook" | tail -3
