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
                                        ("examples/count_in_file/count_in_file.cc", []),
                                        ("examples/query_per_sequence/query_per_sequence.cc", ["-I" + os.path.join(REF, "examples", "query_per_sequence")]),
                                        ("unit_tests/test_hash_counter.cc", ["-I" + os.path.join(REF, "unit_tests")])])
def test_reference_client_sources_compile_unchanged(src, extra, tmp_path):
    path = os.path.join(REF, src)
    if not os.path.exists(path) or shutil.which("g++") is None:
        pytest.skip("reference sources not present")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-c", path, "-o", str(tmp_path / "o.o")] + INC + extra)


def _link_reference_example(src, extra, tmp_path):
    """The reference's example source, unchanged, built into a program against the compat headers + libjfgpu.so.  The read
    side (headers, readers, queries) is host code: these programs run without a GPU."""
    path = os.path.join(REF, src)
    lib = os.path.join(ROOT, "jellyfish_amd", "lib", "libjfgpu.so")
    if not os.path.exists(path) or shutil.which("g++") is None or not os.path.exists(lib):
        pytest.skip("reference sources or the engine library not present")
    exe = str(tmp_path / os.path.basename(src).replace(".cc", ""))
    subprocess.check_call(["g++", "-std=c++17", "-O1", path, "-o", exe] + INC + extra +
                          ["-L" + os.path.dirname(lib), "-l:" + os.path.basename(lib), "-Wl,-rpath," + os.path.dirname(lib), "-pthread"])
    return exe


def _golden_counts(name):
    return {l.split()[0]: int(l.split()[1]) for l in open(os.path.join(GOLD, name + ".dump"))}


def test_reference_query_per_sequence_example_reads_our_and_reference_files(tmp_path):
    """examples/query_per_sequence/query_per_sequence.cc (file_header, mapped_file, binary_query, whole_sequence_parser,
    mer_dna_bloom_counter through jellyfish/*.hpp of the compat directory) on a REFERENCE-written binary/sorted file and on
    a reference-written Bloom counter: per record, the count of every k-mer in order -- equal to what the golden dump says."""
    exe = _link_reference_example("examples/query_per_sequence/query_per_sequence.cc", ["-I" + os.path.join(REF, "examples", "query_per_sequence")], tmp_path)
    man = json.load(open(os.path.join(GOLD, "manifest.json")))
    case = next(c for c in man["cases"] if c["name"] == "reads150_k21C")
    want = _golden_counts("reads150_k21C")
    comp = str.maketrans("ACGT", "TGCA")
    fa = os.path.join(GOLD, case["input"])
    out = subprocess.check_output([exe, os.path.join(GOLD, case["ref_jf"]), fa], text=True).splitlines()
    recs = open(fa).read().split(">")[1:]
    assert len(out) == 2 * len(recs)
    for i, rec in enumerate(recs):
        hdr, seq = rec.split("\n", 1)
        seq = seq.replace("\n", "")
        assert out[2 * i] == ">" + hdr
        exp = []
        for j in range(len(seq) - 20):
            m = seq[j:j + 21].upper()
            if set(m) <= set("ACGT"):
                rc = m.translate(comp)[::-1]
                exp.append(str(want.get(min(m, rc), 0)))
        assert out[2 * i + 1].split() == exp
    # the Bloom counter branch: every k-mer of the file it was made from answers 1 or 2, never 0
    bc = os.path.join(GOLD, "bc_k21C.ref.bc")
    b = man["bloom"][0] if isinstance(man.get("bloom"), list) else None
    src = os.path.join(GOLD, (b or {}).get("input", case["input"]))
    lines = subprocess.check_output([exe, bc, src], text=True).splitlines()
    vals = [v for l in lines[1::2] for v in l.split()]
    assert vals and set(vals) <= {"1", "2"}


def test_reference_count_in_file_example_merges_two_files(tmp_path):
    """examples/count_in_file/count_in_file.cc (binary_reader + mer_heap over file headers): the same reference-written
    file given twice -> every k-mer once, with its count in both columns, in the file's (pos, key) order."""
    exe = _link_reference_example("examples/count_in_file/count_in_file.cc", [], tmp_path)
    jf = os.path.join(GOLD, "reads150_k21C.ref.jf")
    out = subprocess.check_output([exe, jf, jf], text=True).splitlines()
    want = _golden_counts("reads150_k21C")
    got = {}
    for l in out:
        k, a, b = l.split()
        assert a == b and k not in got
        got[k] = int(a)
    assert got == want


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
