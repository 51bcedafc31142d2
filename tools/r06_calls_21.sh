# round 6, call 21: the tile stage on single tiles (JFGPU_TILE_PAIR=0) with 512-thread workgroups (two per CU by registers) and with
# 256-thread workgroups (-DJFGPU_T_BLOCK=256 -DJFGPU_T_QUEUE=3072: four per CU), against the pair kernel (default).  Only T matters here: P2 to 2048
# destinations takes the sort-based kernel.
O=gpurun_out
{
echo "--- default (pairs, 512 threads)"; JFGPU_MATRIX=xs python tools/c2_stage_times.py 2>&1 | grep "^k 21"
echo "--- single tiles, 512 threads"; JFGPU_TILE_PAIR=0 JFGPU_MATRIX=xs python tools/c2_stage_times.py 2>&1 | grep "^k 21"
echo "--- single tiles, 256 threads"; JFGPU_TILE_PAIR=0 JFGPU_MATRIX=xs JFGPU_LIB=jellyfish_amd/lib/libjfgpu_t256.so python tools/c2_stage_times.py 2>&1 | tail -3
} > $O/r06_call21.log 2>&1
cat $O/r06_call21.log
