tools/r06_call.sh p1xs \
 "sh:python tools/c2_stage_times.py" \
 "sh:JFGPU_MATRIX=xs python tools/c2_stage_times.py" \
 "sh:JFGPU_MATRIX=xs JFGPU_LIB=jellyfish_amd/lib/libjfgpu_late.so python tools/c2_stage_times.py" \
 "sh:JFGPU_MATRIX=xs python -m pytest -m gpu -x -q -p no:cacheprovider tests/test_gpu_parity.py -k 'ring_p2_kernels or slot32_equals or single_pass_p1 or comm_item_path or test_count_matches_oracle or high_coverage'" \
 "sh:JFGPU_MATRIX=xs JFGPU_LIB=jellyfish_amd/lib/libjfgpu_late.so python -m pytest -m gpu -x -q -p no:cacheprovider tests/test_gpu_parity.py -k 'ring_p2_kernels or slot32_equals or single_pass_p1'"
