#!/bin/bash
# Builds the host-emulated engine (tests/host/hip_emu): the kernels' sources compiled by g++ for CPU debugging.
# TEST INFRASTRUCTURE: loaded only when a test run sets JFGPU_EMU=1 (tests/conftest.py); never measured, never shipped.
set -e
cd "$(dirname "$0")/../.."
mkdir -p tests/host/_build
# up to date?  (two test modules ask for the build in one run)
if [ -f tests/host/_build/libjfgpu_emu.so ] && [ -f tests/host/_build/jellyfish-amd-emu ] && \
   [ -z "$(find jellyfish_amd/csrc jellyfish_amd/cli jellyfish_amd/include include tests/host/hip_emu tests/host/build_emu.sh -type f -newer tests/host/_build/jellyfish-amd-emu -print -quit)" ]; then
  exit 0
fi
g++ -std=c++17 -O2 -g -x c++ -DJFGPU_EMU -Itests/host/hip_emu -fPIC -shared -pthread \
    -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result -Wno-unknown-pragmas -Wno-sign-compare -Wno-unused-but-set-variable -Wno-unused-variable \
    -o tests/host/_build/libjfgpu_emu.so jellyfish_amd/csrc/jfgpu.hip
# the CLI against the emulated engine (JFGPU_CLI=tests/host/_build/jellyfish-amd-emu for tests/test_cli_gpu.py)
g++ -O2 -std=c++17 -Iinclude -Ijellyfish_amd/include -o tests/host/_build/jellyfish-amd-emu jellyfish_amd/cli/jellyfish_amd.cc \
    -Ltests/host/_build -ljfgpu_emu -Wl,-rpath,'$ORIGIN' -pthread
