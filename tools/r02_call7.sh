#!/bin/bash
mkdir -p gpurun_out
{
  for v in 0 1; do
    echo "== tile phase clocks, JFGPU_SLOT64=$v"
    JFGPU_LIB=$PWD/jellyfish_amd/lib/libjfgpu_tileprof.so JFGPU_SLOT64=$v timeout 600 python bench.py --no-cpu-baseline --no-extras --repeats 1 --warmup 1 2> gpurun_out/r02_tp$v.err | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('value', d['value'], {k:(x['ms'],x['launches']) for k,x in d['kernels'].items()})"
    grep "tile prof" gpurun_out/r02_tp$v.err
  done
} > gpurun_out/r02_call7.log 2>&1
cat gpurun_out/r02_call7.log | cut -c1-600
