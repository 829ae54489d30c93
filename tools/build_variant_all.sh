#!/bin/bash
# usage: tools/build_variant_all.sh <name> <-Dflags...> — the WHOLE library recompiled under extra flags, as
# zignal_amd/variants/libzignal_hip_<name>.so (git-ignored, travels to the GPU box; pick it with ZIGNAL_HIP_LIBRARY=<path>).
set -e
name=$1; shift
cd "$(dirname "$0")/../zignal_amd/csrc"
mkdir -p ../variants build/var_$name
ls *.hip *.cpp | xargs -P 8 -I{} sh -c "f={}; x=''; case \$f in *.cpp) x='-x hip';; esac; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -Wno-unused-function \$( [ \$f = convert.hip ] && echo -fno-slp-vectorize ) $* \$x -c \$f -o build/var_$name/\$f.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/libzignal_hip_$name.so build/var_$name/*.o -lz
echo built zignal_amd/variants/libzignal_hip_$name.so
