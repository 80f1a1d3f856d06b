#!/bin/bash
mkdir -p gpurun_out/r5
{ python3 scripts/host_call_times.py 216 216; python3 scripts/host_call_times.py 216 27; } > gpurun_out/r5/v.log 2>&1
grep -v amdgpu.ids gpurun_out/r5/v.log
