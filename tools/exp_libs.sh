#!/bin/bash
# stage times of config 2's job for a list of side builds: tools/exp_libs.sh e1 e2 ...   (jellyfish_amd/lib/libjfgpu_<name>.so)
cd "$(dirname "$0")/.."
for v in "$@"; do
  echo "--- $v"
  JFGPU_LIB=jellyfish_amd/lib/libjfgpu_$v.so python tools/c2_stage_times.py 2>&1 | grep -E "^k 21|T:" | tail -2
done
