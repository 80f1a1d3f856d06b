#!/bin/bash
# CPU-only sanitizer pass (ASan + UBSan) over the host generators and the oracle.
set -e
here=$(cd "$(dirname "$0")" && pwd); root=$(cd "$here/../.." && pwd)
out=$(mktemp -d)
g++ -std=c++17 -g -O1 -fsanitize=address,undefined -fno-omit-frame-pointer -fopenmp -D__HIP_PLATFORM_AMD__ \
    -I"$root/include" -I/opt/rocm/include -I"$root/oracle" -I"$root/thunderbolt.jl_amd/csrc" \
    "$here/host_driver.cpp" "$root/thunderbolt.jl_amd/csrc/tb_hostgen.cpp" -x c "$root/oracle/tb_oracle.c" -lm -o "$out/drv"
ASAN_OPTIONS=detect_leaks=1 UBSAN_OPTIONS=halt_on_error=1 "$out/drv"
rm -rf "$out"
