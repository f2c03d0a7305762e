#!/bin/bash
# experiment builds of the library (selected with JF_LIB=...): tools/build_exp.sh <name> <-D flags...>
set -e
name=$1; shift
cd "$(dirname "$0")/.." && mkdir -p tools/exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Iinclude -Ijacobiforcing_amd/csrc "$@" jacobiforcing_amd/csrc/*.hip -o tools/exp/libjf_exp_$name.so
echo built tools/exp/libjf_exp_$name.so
