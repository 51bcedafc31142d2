#!/bin/bash
# Side builds of the engine for A/B runs on the GPU box: tools/side_build.sh <name> [-DFLAG ...]  ->  jellyfish_amd/lib/libjfgpu_<name>.so
# (loaded with JFGPU_LIB=...; tools/exp_libs.sh and tools/ab_bench.sh take the names).  Several can be built in parallel.
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p jellyfish_amd/lib
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-value -Wno-unused-result "$@" -shared \
  -o jellyfish_amd/lib/libjfgpu_$name.so jellyfish_amd/csrc/jfgpu.hip -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib 2>&1 | grep -E "error" 
ls -la jellyfish_amd/lib/libjfgpu_$name.so
