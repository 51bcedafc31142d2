#!/bin/bash
# The product library and the -DJFGPU_PHASE_PROF side build (phase clocks of the partition kernels; tools/ab_bench.sh with JFGPU_LIB=jellyfish_amd/lib/libjfgpu_phaseprof.so).
cd "$(dirname "$0")/.."
make engine 2>&1 | grep -E "error|Error" 
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-value -Wno-unused-result -DJFGPU_PHASE_PROF -shared \
  -o jellyfish_amd/lib/libjfgpu_phaseprof.so jellyfish_amd/csrc/jfgpu.hip -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib 2>&1 | grep -E "error"
ls -la jellyfish_amd/lib/
