#!/bin/bash
# usage: tools/build_variant.sh <name> <file.hip> <-Dflags...> — the library with ONE source recompiled under extra flags, as
# zignal_amd/variants/libzignal_hip_<name>.so (git-ignored, travels to the GPU box; pick it with ZIGNAL_HIP_LIBRARY=<path>). For A/B timing of
# experiment macros ("pieces removed") without touching the product build.
set -e
name=$1; src=$2; shift 2
cd "$(dirname "$0")/../zignal_amd/csrc"
mkdir -p ../variants build/var_$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -Wall -Wno-unused-function "$@" -c $src -o build/var_$name/$src.o
objs=$(ls build/*.o | grep -v "build/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/libzignal_hip_$name.so $objs build/var_$name/$src.o -lz
echo built zignal_amd/variants/libzignal_hip_$name.so
