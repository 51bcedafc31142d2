"""The oracle must BE the reference before anything is compared against it (CPU only).

1. oracle/jf_oracle.c (plain-C restatement) reproduces the committed golden fixtures,
   which are outputs of the reference's own classes (oracle/gen_golden.py).
2. Where oracle/_ref exists (this container), the restatement is also fuzzed against
   the reference on parser edge cases, and oracle/_ref is re-pinned to the reference's
   golden md5s (tests/parallel_hashing.sh:7-19)."""
import hashlib
import json
import os
import random
import subprocess

import numpy as np
import pytest

import oracle_lib as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MANIFEST = json.load(open(os.path.join(GOLD, "manifest.json")))


def restated_dump(path, k, canonical):
    seq = O.parse_file(open(path, "rb").read())
    keys, cnt = O.count(seq, k, canonical)
    return sorted("%s %d" % (O.to_str(keys[i], k), cnt[i]) for i in range(len(keys)))


@pytest.mark.parametrize("case", MANIFEST["cases"], ids=lambda c: c["name"])
def test_restatement_reproduces_reference_fixture(case):
    got = restated_dump(os.path.join(GOLD, case["input"]), case["k"], case["canonical"])
    exp = open(os.path.join(GOLD, case["name"] + ".dump")).read().splitlines()
    assert got == exp
    # stats/histo derived from the same map match the reference's text outputs
    counts = [int(l.split()[1]) for l in got]
    stats = "Unique:    %d\nDistinct:  %d\nTotal:     %d\nMax_count: %d\n" % (
        sum(c == 1 for c in counts), len(counts), sum(counts), max(counts) if counts else 0)
    assert stats == open(os.path.join(GOLD, case["name"] + ".stats")).read()
    h = {}
    for c in counts:
        h[c] = h.get(c, 0) + 1
    histo = "".join("%d %d\n" % (c, h[c]) for c in sorted(h))
    assert histo == open(os.path.join(GOLD, case["name"] + ".histo")).read()


def read_jf(path):
    """Independent decoder of binary/sorted: header JSON + fixed records."""
    data = open(path, "rb").read()
    hlen = int(data[:9])
    header = json.loads(data[9:9 + hlen].rstrip(b"\0"))
    body = data[9 + hlen:]
    kb = (header["key_len"] + 7) // 8
    rec = kb + header["counter_len"]
    assert len(body) % rec == 0 and (9 + hlen) % 8 == 0
    recs = np.frombuffer(body, dtype=np.uint8).reshape(-1, rec)
    return header, recs, kb


def test_matrix_times_matches_reference_file_order():
    """The reference wrote reads150_k21C.ref.jf sorted by (matrix1 * key & (size-1), key)
    (mer_heap.hpp:26-30): recomputing pos with the restated times() must find it sorted."""
    header, recs, kb = read_jf(os.path.join(GOLD, "reads150_k21C.ref.jf"))
    keys = np.zeros(len(recs), dtype=np.uint64)
    for b in range(kb):
        keys |= recs[:, b].astype(np.uint64) << np.uint64(8 * b)
    m = header["matrix1"]
    cols = None if m["identity"] else np.array(m["columns"], dtype=np.uint64)
    pos = O.matrix_times(cols, m["r"], m["c"], keys) & np.uint64(header["size"] - 1)
    pk = list(zip(pos.tolist(), keys.tolist()))
    assert pk == sorted(pk) and len(set(pk)) == len(pk)
    assert header["format"] == "binary/sorted" and header["canonical"] is True
    exp = open(os.path.join(GOLD, "reads150_k21C.dump")).read().splitlines()
    cnt = recs[:, kb].astype(np.uint64)
    for b in range(1, header["counter_len"]):
        cnt |= recs[:, kb + b].astype(np.uint64) << np.uint64(8 * b)
    got = sorted("%s %d" % (O.to_str(np.array([keys[i]]), 21), cnt[i]) for i in range(len(keys)))
    assert got == exp


def test_codes_table():
    """mer_dna.hpp:38-55 / unit_tests/test_mer_dna.cc:97-124: only ACGTacgt are bases; U is not T."""
    L = O.lib()
    good = {c: i for i, pair in enumerate(("Aa", "Cc", "Gg", "Tt")) for c in pair}
    for c in range(256):
        code = L.jfo_code(c)
        if chr(c) in good:
            assert code == good[chr(c)]
        else:
            assert code < 0
    assert L.jfo_code(ord("\n")) == -2 and L.jfo_code(ord("N")) == -1 and L.jfo_code(ord("U")) == -3


def test_revcomp_and_canonical_vectors():
    """unit_tests/test_mer_dna.cc:505-529 style checks on fixed strings."""
    def rc(s):
        return s[::-1].translate(str.maketrans("ACGT", "TGCA"))
    rng = random.Random(3)
    for k in (1, 2, 7, 31, 32, 33, 64, 65, 100):
        for _ in range(20):
            s = "".join(rng.choice("ACGT") for _ in range(k))
            key = O.from_str(s, k)
            out = np.zeros_like(key)
            O.lib().jfo_revcomp(key.ctypes.data, out.ctypes.data, k)
            assert O.to_str(out, k) == rc(s)
            got = O.extract(s.encode(), k, True)
            assert O.to_str(got[0], k) == min(s, rc(s))


def test_bloom_counter_restatement_saturates_at_two():
    """bloom_counter2.hpp:56-107: digits are base 3, increments saturate at 2, insert returns
    the minimum previous digit; final state is order independent."""
    m, nh = 1000, 7
    data = np.zeros((m + 4) // 5, dtype=np.uint8)
    L = O.lib()
    assert L.jfo_bc_check(data.ctypes.data, m, nh, 12345, 6789) == 0
    assert L.jfo_bc_insert(data.ctypes.data, m, nh, 12345, 6789) == 0
    assert L.jfo_bc_check(data.ctypes.data, m, nh, 12345, 6789) == 1
    assert L.jfo_bc_insert(data.ctypes.data, m, nh, 12345, 6789) == 1
    assert L.jfo_bc_insert(data.ctypes.data, m, nh, 12345, 6789) == 2
    assert L.jfo_bc_check(data.ctypes.data, m, nh, 12345, 6789) == 2
    assert data.max() <= 242


needs_ref = pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built (needs /root/reference)")


@needs_ref
def test_restatement_fuzz_against_reference(tmp_path):
    rng = random.Random(1)

    def rnd(n, alphabet="ACGT"):
        return "".join(rng.choice(alphabet) for _ in range(n))
    cases = {
        "fasta_multi": "".join(">r%d desc\n%s\n%s\n" % (i, rnd(70), rnd(rng.randint(1, 70))) for i in range(50)),
        "fasta_lower_N": "".join(">r%d\n%s\n" % (i, rnd(300, "ACGTacgtACGTacgtNRY-")) for i in range(30)),
        "fasta_blank": ">a\n\nACGTACGTACGTACGTACGTAAAA\n\n\nCCCCGGGGTTTTACACACACACAC\n>b\n>c\nACGTTTTTTTTTTTTTTTTTTTGGGGGGGGG",
    }
    cases["fasta_crlf"] = cases["fasta_multi"].replace("\n", "\r\n")
    fq = ""
    for i in range(40):
        s = rnd(rng.randint(30, 120))
        fq += "@r%d\n%s\n+\n%s\n" % (i, s, "".join(rng.choice("IJ@+>") for _ in s))
    cases["fastq"] = fq
    fq2 = ""
    for i in range(10):
        q = "".join(rng.choice("IJ@+>") for _ in range(100))
        fq2 += "@r%d\n%s\n%s\n+r%d\n%s\n%s\n" % (i, rnd(60), rnd(40), i, q[:60], q[60:])
    cases["fastq_multiline"] = fq2
    for name, txt in cases.items():
        p = tmp_path / (name + ".fa")
        p.write_text(txt)
        for k, can in [(5, True), (21, True), (21, False), (31, True), (33, True), (63, True), (100, False)]:
            ref, _ = O.ref_count_dump(str(p), k, 100000, canonical=can, workdir=str(tmp_path))
            assert restated_dump(str(p), k, can) == ref, (name, k, can)


@needs_ref
def test_oracle_ref_reproduces_reference_golden_md5(tmp_path):
    """tests/parallel_hashing.sh:7-19 on the reference's own seeded inputs."""
    g = MANIFEST["reference_md5"]
    d = str(tmp_path)
    subprocess.check_call([O.REF_GEN, "-o", "seq10m"] + g["seq10m"], cwd=d)
    subprocess.check_call([O.REF_GEN, "-o", "seq1m"] + g["seq1m"], cwd=d)

    def md5(cmd, post=None):
        out = subprocess.check_output(cmd, cwd=d)
        if post:
            out = post(out)
        return hashlib.md5(out).hexdigest()
    subprocess.check_call([O.REF_JF, "count", "-t", "4", "-o", "m15.jf", "-s", "2M", "-C", "-m", "15", "seq10m.fa"], cwd=d)
    assert md5([O.REF_JF, "histo", "m15.jf"]) == g["m15_s2M.histo"]
    assert md5([O.REF_JF, "stats", "m15.jf"]) == g["m15.stats"]
    subprocess.check_call([O.REF_JF, "count", "-m", "40", "-t", "4", "-o", "bin.jf", "-s", "2M", "seq1m_0.fa"], cwd=d)
    assert md5([O.REF_JF, "dump", "-c", "bin.jf"], lambda o: b"".join(sorted(o.splitlines(True)))) == g["binary.dump"]
    assert md5([O.REF_JF, "histo", "bin.jf"]) == g["binary.histo"]
    assert md5([O.REF_JF, "stats", "bin.jf"]) == g["binary.stats"]
    subprocess.check_call([O.REF_JF, "count", "-t", "4", "-o", "lu.jf", "-s", "2M", "-C", "-m", "15", "-L", "2", "-U", "3", "seq10m.fa"], cwd=d)
    assert md5([O.REF_JF, "histo", "lu.jf"]) == g["m15_s2M_L2_U3.histo"]


def read_bc(path):
    data = open(path, "rb").read()
    hlen = int(data[:9])
    header = json.loads(data[9:9 + hlen].rstrip(b"\0"))
    return header, np.frombuffer(data[9 + hlen:], dtype=np.uint8)


@pytest.mark.parametrize("case", MANIFEST["bloom"], ids=lambda c: c["name"])
def test_bloom_restatement_reproduces_reference_file(case):
    """The reference wrote <name>.ref.bc (bc_main.cc:109-148).  Replaying row a14 of SURVEY 8(a)
    (cells (h0 % m + i (h1 % m)) % m, saturating base-3 digits) with the file's own matrices must
    give the byte-identical body, and `count --bc` (check > 1) the reference's filtered dump."""
    header, body = read_bc(os.path.join(GOLD, case["ref_bc"]))
    k, can = case["k"], case["canonical"]
    assert header["format"] == "bloomcounter" and header["key_len"] == 2 * k
    m, nh = header["size"], header["nb_hashes"]
    assert m == 9000 * 14 and nh == 10 and len(body) == (m + 4) // 5       # opt_m / opt_k at fpr 0.001
    m1 = np.array(header["matrix1"]["columns"], dtype=np.uint64)
    m2 = np.array(header["matrix2"]["columns"], dtype=np.uint64)
    seq = O.parse_file(open(os.path.join(GOLD, case["input"]), "rb").read())
    kmers = O.extract(seq, k, can)
    h0 = O.matrix_times(m1, 64, 2 * k, kmers)
    h1 = O.matrix_times(m2, 64, 2 * k, kmers)
    data = np.zeros(len(body), dtype=np.uint8)
    L = O.lib()
    for a, b in zip(h0.tolist(), h1.tolist()):
        L.jfo_bc_insert(data.ctypes.data, m, nh, a, b)
    assert (data == body).all()
    keys, cnt = O.count(seq, k, can)
    kh0 = O.matrix_times(m1, 64, 2 * k, keys)
    kh1 = O.matrix_times(m2, 64, 2 * k, keys)
    kept = sorted("%s %d" % (O.to_str(keys[i], k), cnt[i]) for i in range(len(keys))
                  if L.jfo_bc_check(data.ctypes.data, m, nh, int(kh0[i]), int(kh1[i])) > 1)
    assert kept == open(os.path.join(GOLD, case["name"] + ".filtered.dump")).read().splitlines()


def _digest_lines(txt):
    return tuple(int(l.split()[1]) for l in txt.strip().splitlines())


@needs_ref
@pytest.mark.parametrize("k,can", [(21, True), (40, False), (63, True), (100, True)])
def test_content_digest_three_ways(tmp_path, k, can):
    """The at-scale parity checksum (include/jfgpu.h, jfgpu_digest): the reference driver computes it from the reference's
    in-memory table (`count --digest`, large_hash_array iterators) and from a file (`digest`); both equal the numpy
    restatement applied to the oracle's own counts.  One-, two- and four-word keys."""
    from jellyfish_amd import capi
    fa = os.path.join(GOLD, "reads150_s42.fa")
    d = str(tmp_path)
    args = [O.REF_JF, "count", "-m", str(k), "-s", "64k", "-t", "3", "-o", "x.jf", "--digest", "mem.txt", fa]
    if can:
        args.insert(2, "-C")
    subprocess.check_call(args, cwd=d)
    mem = _digest_lines(open(os.path.join(d, "mem.txt")).read())
    fil = _digest_lines(subprocess.check_output([O.REF_JF, "digest", "x.jf"], cwd=d).decode())
    keys, cnt = O.count(O.parse_file(open(fa, "rb").read()), k, can)
    mine = capi.digest_of(keys, cnt)
    assert mem == fil == mine
    # with a count filter
    args[args.index("mem.txt")] = "lu.txt"
    subprocess.check_call(args[:2] + ["-L", "2", "-U", "3"] + args[2:], cwd=d)
    sel = (cnt >= 2) & (cnt <= 3)
    assert _digest_lines(open(os.path.join(d, "lu.txt")).read()) == capi.digest_of(keys[sel], cnt[sel])


@needs_ref
def test_config1_recorded_md5s(tmp_path):
    """BASELINE configs[0]: `generate_sequence -s 42 -r 150 -o reads150 10000000` is the 10 MB
    FASTA whose md5 the survey recorded, and `jellyfish count -m 21 -C -s 16M -t 1` + `histo` on it gives the recorded
    histogram md5 -- through the reference's classes (oracle/_ref) and, same lines, through the C restatement."""
    d = str(tmp_path)
    subprocess.check_call([O.REF_GEN, "-s", "42", "-r", "150", "-o", "reads150", "10000000"], cwd=d)
    fa = os.path.join(d, "reads150.fa")
    data = open(fa, "rb").read()
    assert len(data) == 10922231 and hashlib.md5(data).hexdigest() == "e539471302c480c328f8908dad38ff67"
    subprocess.check_call([O.REF_JF, "count", "-m", "21", "-C", "-s", "16M", "-t", "1", "-o", "c1.jf", "reads150.fa"], cwd=d)
    histo = subprocess.check_output([O.REF_JF, "histo", "c1.jf"], cwd=d)
    assert histo == b"1 8666626\n2 17\n" and hashlib.md5(histo).hexdigest() == "08762de4d79a53b64b58517437425c3e"
    keys, cnt = O.count(O.parse_file(data), 21, True)
    assert int(cnt.sum()) == 8666660 and len(keys) == 8666643
    vals, n = np.unique(cnt, return_counts=True)
    assert "".join("%d %d\n" % (v, c) for v, c in zip(vals.tolist(), n.tolist())).encode() == histo
