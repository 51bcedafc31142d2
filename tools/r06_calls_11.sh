# round 6, call 11: P1b through rings of 256 bytes (p1_bloom_ring_kernel): parity (Bloom suite, forced-ring cases), then config 3 A/B;
# the xs matrix files read by the reference's reader; k = 31 plain count under both matrix families
O=gpurun_out
{
timeout 1500 python -m pytest tests/test_gpu_bloom.py -x -q 2>&1 | tail -15
timeout 900 python -m pytest tests/test_cli_gpu.py -x -q -k "xorshift or too_small" 2>&1 | tail -15
} > $O/r06_call11_tests.log 2>&1
tail -30 $O/r06_call11_tests.log
{
bash tools/ab_bench.sh "C3ring::--config C3" "C3sort:JFGPU_BLOOM_P1_RING=0:--config C3" "C3ringbar2:JFGPU_LIB=jellyfish_amd/lib/libjfgpu_bar2.so:--config C3" "K31xs::--config K31" "K31ref::--config K31 --matrix reference" "C2xs::" 
} > $O/r06_call11_ab.log 2>&1
cat $O/r06_call11_ab.log
