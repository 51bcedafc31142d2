#!/bin/bash
# Runs a command (default: the -m gpu tests given as arguments) against the host-emulated engine (tests/host/hip_emu).
#   tests/host/run_emu.sh python -m pytest tests/test_gpu_bloom.py -m gpu -x -q
# Debugging aid only: a pass here says the kernels' logic is right, not that they are correct or fast on the device.
here="$(cd "$(dirname "$0")" && pwd)"
[ -f "$here/_build/libjfgpu_emu.so" ] || "$here/build_emu.sh" || exit 1
export JFGPU_LIB="$here/_build/libjfgpu_emu.so" JFGPU_CLI="$here/_build/jellyfish-amd-emu"
exec "$@"
