"""CPU run of the kernels' SOURCE (not -m gpu): jellyfish_amd/csrc compiled by g++ against tests/host/hip_emu, a host
stand-in for the HIP execution model (work-items as fibers with real barrier / shuffle rendezvous), then a selection of
the -m gpu parity tests executed against that library in a subprocess.

What this is: a logic check of the device code on every CPU test run -- partition placement, LDS tile protocols, the
multi-GPU exchange bookkeeping at world sizes 2 and 4 (local transport), the Bloom segment path.  What it is not: a CPU
fallback (the library is test infrastructure under tests/host/_build, loaded only here via JFGPU_LIB; the product
library has no CPU path) or evidence about races, memory ordering or speed -- those only the GPU run can give."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "host", "_build")

SELECTION = [
    "tests/test_gpu_parity.py::test_comm_local_transport_equals_single_table",
    "tests/test_gpu_parity.py::test_comm_item_path_equals_single_table[4-None]",
    "tests/test_gpu_parity.py::test_comm_item_path_equals_single_table[2-3]",
    "tests/test_gpu_parity.py::test_p2_variants_of_32bit_slots_give_the_same_table[single_pass]",
    "tests/test_gpu_parity.py::test_p2_variants_of_32bit_slots_give_the_same_table[tiles]",
    "tests/test_cli_gpu.py::test_file_parts_cover_the_file_exactly_once",
    "tests/test_cli_gpu.py::test_pipes_are_read_in_pieces_of_whole_records",
    "tests/test_gpu_wide.py::test_dump_of_saturated_count_fields_over_all_ones_tags",
    "tests/test_compat.py::test_hash_counter_check_like_the_reference_unit_test",
    "tests/test_gpu_parity.py::test_ragged_lengths",
    "tests/test_gpu_parity.py::test_add_key_vals_loads_pairs",
    "tests/test_gpu_parity.py::test_spill_mode_add_keys_and_tiny_pieces",
    "tests/test_gpu_parity.py::test_add_keys_batch_larger_than_the_table_grows_in_order",
    "tests/test_gpu_bloom.py::test_partitioned_insert_equals_direct_and_oracle",
    "tests/test_gpu_wide.py::test_keys_of_three_and_four_words",
    "tests/test_gpu_wide.py::test_wide_partitioned_path_equals_direct_and_oracle[33-65536]",
    "tests/test_gpu_wide.py::test_wide_partitioned_path_equals_direct_and_oracle[40-1048576]",
    "tests/test_gpu_parity.py::test_shards_grow_together[2-0]",
    "tests/test_gpu_wide.py::test_two_word_shards_grow_together[40-2]",
    "tests/test_gpu_parity.py::test_prime_and_update_over_shards[2]",
    "tests/test_gpu_bloom.py::test_sharded_count_with_a_bloom_counter_equals_the_single_table[21-2-2]",
    "tests/test_gpu_bloom.py::test_count_bc_with_the_cache_of_admitted_kmers[bc_k21C-2]",
    "tests/test_gpu_bloom.py::test_bloom_counters_of_the_ranks_merge_into_the_counter_of_the_whole_input[2]",
]


@pytest.fixture(scope="module")
def emu_lib():
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    subprocess.check_call([os.path.join(ROOT, "tests", "host", "build_emu.sh")])
    lib = os.path.join(BUILD, "libjfgpu_emu.so")
    assert os.path.exists(lib)
    return lib


def test_device_sources_pass_their_parity_tests_on_the_host_emulation(emu_lib):
    env = dict(os.environ, JFGPU_LIB=emu_lib, JFGPU_CLI=os.path.join(BUILD, "jellyfish-amd-emu"), JFGPU_EMU_THREADS="4")
    r = subprocess.run([sys.executable, "-m", "pytest", "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider"] + SELECTION,
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout
