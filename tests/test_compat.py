"""The source-compatible include directory (jellyfish_amd/compat/jellyfish/*.hpp, namespace jellyfish): programs written
against the reference's C++ API build against the engine unchanged.

CPU part (not -m gpu): the reference's OWN client sources -- examples/jf_count_dump/jf_count_dump.cc and
unit_tests/test_hash_counter.cc (with the gtest header it ships) -- are compiled, where they lie under /root/reference,
against -Ijellyfish_amd/compat; skipped where the reference is absent (the GPU box).  GPU part: two programs of this repo
that use the same API the same way (tests/compat/) are built, linked with the engine and run: the count/dump example's
output must be the golden dump, the hash-counter check must pass."""
import json
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"
INC = ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "jellyfish_amd", "compat"), "-I" + os.path.join(ROOT, "jellyfish_amd", "include")]


@pytest.mark.parametrize("src,extra", [("examples/jf_count_dump/jf_count_dump.cc", []),
                                        ("unit_tests/test_hash_counter.cc", ["-I" + os.path.join(REF, "unit_tests")])])
def test_reference_client_sources_compile_unchanged(src, extra, tmp_path):
    path = os.path.join(REF, src)
    if not os.path.exists(path) or shutil.which("g++") is None:
        pytest.skip("reference sources not present")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-c", path, "-o", str(tmp_path / "o.o")] + INC + extra)


def _build(name, tmp_path):
    lib = os.environ.get("JFGPU_LIB") or os.path.join(ROOT, "jellyfish_amd", "lib", "libjfgpu.so")
    exe = str(tmp_path / name)
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", os.path.join(ROOT, "tests", "compat", name + ".cc"), "-o", exe] + INC +
                          ["-L" + os.path.dirname(lib), "-l:" + os.path.basename(lib), "-Wl,-rpath," + os.path.dirname(lib), "-pthread"])
    return exe


@pytest.mark.gpu
@pytest.mark.parametrize("name,threads", [("reads150_k21C", 4), ("reads150_k32", 1), ("reads150_k63C", 3)])
def test_count_dump_program_written_against_the_reference_api(gpu, tmp_path, name, threads):
    case = next(c for c in json.load(open(os.path.join(GOLD, "manifest.json")))["cases"] if c["name"] == name)
    exe = _build("count_dump", tmp_path)
    r = subprocess.run([exe, str(case["k"]), "1" if case["canonical"] else "0", str(threads), os.path.join(GOLD, case["input"])],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    golden = open(os.path.join(GOLD, name + ".dump")).read().splitlines()
    assert sorted(r.stdout.splitlines()) == golden
    assert "lookups_ok 500" in r.stderr


@pytest.mark.gpu
def test_hash_counter_check_like_the_reference_unit_test(gpu, tmp_path):
    exe = _build("hash_counter_check", tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip() == "OK", r.stderr[-2000:]
