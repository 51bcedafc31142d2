tools/r06_call.sh p1stage \
 "sh:JFGPU_MATRIX=xs python tools/c2_stage_times.py" \
 "sh:python tools/c2_stage_times.py" \
 "sh:JFGPU_MATRIX=xs JFGPU_LIB=jellyfish_amd/lib/libjfgpu_oldstage.so python tools/c2_stage_times.py" \
 "sh:JFGPU_MATRIX=xs JFGPU_LIB=jellyfish_amd/lib/libjfgpu_phaseprof.so python tools/c2_stage_times.py" \
 "sh:JFGPU_MATRIX=xs python -m pytest -m gpu -x -q -p no:cacheprovider tests/test_gpu_parity.py -k 'ring_p2_kernels or slot32_equals or single_pass_p1 or comm_item_path or high_coverage or ragged or unaligned or calls_do_not'" \
 "sh:python -m pytest -m gpu -x -q -p no:cacheprovider tests/test_gpu_parity.py -k 'ring_p2_kernels or slot32_equals or single_pass_p1 or ragged or unaligned or calls_do_not'"
cat gpurun_out/r06_p1stage.log | grep -E "^k 21|phase prof" 
