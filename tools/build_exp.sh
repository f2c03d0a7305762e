#!/bin/bash
# experiment builds of the library (selected with JF_LIB=...): tools/build_exp.sh <name> <-D flags...>
set -e
name=$1; shift
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Iinclude -Ijacobiforcing_amd/csrc "$@" jacobiforcing_amd/csrc/*.hip -o tools/libjf_exp_$name.so
echo built tools/libjf_exp_$name.so
