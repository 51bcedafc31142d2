# round 6, call 18: buckets of eight slots (kBucketBits = 3: the tile kernel generalised over the bucket size) against buckets of four
O=gpurun_out
{
for rep in 1 2; do
echo "--- buckets of eight (xs)"; JFGPU_MATRIX=xs python tools/c2_stage_times.py 2>&1 | grep "^k 21"
echo "--- buckets of four (xs)"; JFGPU_MATRIX=xs JFGPU_LIB=jellyfish_amd/lib/libjfgpu_b4.so python tools/c2_stage_times.py 2>&1 | grep "^k 21"
done
bash tools/ab_bench.sh "G8::--dist G --repeats 2" "G4:JFGPU_LIB=jellyfish_amd/lib/libjfgpu_b4.so:--dist G --repeats 2" "K31b8::--config K31" "K31b4:JFGPU_LIB=jellyfish_amd/lib/libjfgpu_b4.so:--config K31" "C3b8::--config C3 --repeats 2"
} > $O/r06_call18.log 2>&1
cat $O/r06_call18.log
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > $O/r06_call18_suite.log
cat $O/r06_call18_suite.log
