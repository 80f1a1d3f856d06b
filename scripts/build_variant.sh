#!/bin/bash
# scripts/build_variant.sh NAME FILE.hip "-DFLAG=… …": libtbhip_NAME.so = the current build with ONE translation unit recompiled under extra flags
# (same-box A/B of compile-time choices: TB_LIBTBHIP=thunderbolt.jl_amd/libtbhip_NAME.so selects it in the Python loader)
set -e
cd "$(dirname "$0")/../thunderbolt.jl_amd/csrc"
name=$1; src=$2; flags=$3
mkdir -p build_$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=fast -I../../include -I/opt/rocm/include -Wall -Wno-unused-function $flags -c $src -o build_$name/${src%.hip}.o
objs=$(ls build/*.o | grep -v "/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libtbhip_$name.so $objs build_$name/${src%.hip}.o -lgomp -ldl -Wl,-rpath,/opt/rocm/lib
echo built ../libtbhip_$name.so
