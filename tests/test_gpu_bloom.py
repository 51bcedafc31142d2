"""BASELINE config 3 on the GPU (-m gpu): Bloom-counter first pass (`jellyfish bc`) and `count --bc`,
bit/byte-exact against the reference's golden files and the oracle restatement."""
import json
import os
import random
import subprocess

import numpy as np
import pytest

import oracle_lib as O
from test_oracle import read_bc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
MANIFEST = json.load(open(os.path.join(GOLD, "manifest.json")))


@pytest.mark.parametrize("case", MANIFEST["bloom"], ids=lambda c: c["name"])
def test_bloom_bytes_identical_to_reference(gpu, case):
    header, body = read_bc(os.path.join(GOLD, case["ref_bc"]))
    k, can = case["k"], case["canonical"]
    m1 = np.array(header["matrix1"]["columns"], dtype=np.uint64)
    m2 = np.array(header["matrix2"]["columns"], dtype=np.uint64)
    seq = O.parse_file(open(os.path.join(GOLD, case["input"]), "rb").read())
    with gpu.Bloom(k, header["size"], header["nb_hashes"], canonical=can, matrix1=m1, matrix2=m2) as b:
        assert b.nb_bytes == len(body)
        b.insert_ascii(seq)
        assert b.sync() == len(O.extract(seq, k, can))
        assert (b.read() == body).all()                       # same file body as jellyfish bc wrote
        # check() on encoded k-mers: everything inserted reads back >= 1; twice-seen k-mers read 2
        keys, cnt = O.count(seq, k, can)
        chk = b.keys(keys[:, 0])
        assert ((chk == 2) | (cnt < 2)).all() and (chk >= 1).all()
        # count --bc: only k-mers the filter saw at least twice are admitted (count_main.cc:115-118)
        golden = open(os.path.join(GOLD, case["name"] + ".filtered.dump")).read().splitlines()
        for mode in (1, 2):
            with gpu.Table(k, 1 << 16, canonical=can) as t:
                t.set_mode(mode)
                t.attach_bloom(b)
                t.count_ascii(seq)
                t.sync()
                recs = t.dump_records()
                kk, cc = gpu.decode_records(recs, k, 4)
                got = sorted("%s %d" % (O.to_str(np.array([a], dtype=np.uint64), k), c) for a, c in zip(kk.tolist(), cc.tolist()))
                assert got == golden, mode
                assert t.stats().mers_fed == len(O.extract(seq, k, can))     # ++count for every mer, filtered or not
                t.attach_bloom(None)


@pytest.mark.parametrize("log2_sets", ["2", "12"])
@pytest.mark.parametrize("case", MANIFEST["bloom"], ids=lambda c: c["name"])
def test_count_bc_with_the_cache_of_admitted_kmers(gpu, monkeypatch, case, log2_sets):
    """count --bc with the cache of admitted k-mers forced on (JFGPU_BLOOM_CACHE=1; the engine turns it on by itself when a
    pass admits most of its windows -- high-coverage input): a hit answers instead of the counter's ten cells, so the
    filtered table must still be the reference's golden dump (count_main.cc:109-119, bloom_counter2.hpp:109-142), with a
    cache of four sets (every insert evicts) and with one that holds everything, on both insert paths, the input fed
    twice so that the second feed hits."""
    monkeypatch.setenv("JFGPU_BLOOM_CACHE", "1")
    monkeypatch.setenv("JFGPU_BLOOM_CACHE_LOG2", log2_sets)
    header, body = read_bc(os.path.join(GOLD, case["ref_bc"]))
    k, can = case["k"], case["canonical"]
    m1 = np.array(header["matrix1"]["columns"], dtype=np.uint64)
    m2 = np.array(header["matrix2"]["columns"], dtype=np.uint64)
    seq = O.parse_file(open(os.path.join(GOLD, case["input"]), "rb").read())
    golden = dict((l.split()[0], int(l.split()[1])) for l in open(os.path.join(GOLD, case["name"] + ".filtered.dump")).read().splitlines())
    with gpu.Bloom(k, header["size"], header["nb_hashes"], canonical=can, matrix1=m1, matrix2=m2) as b:
        b.load(body)
        for mode in (1, 2):
            with gpu.Table(k, 1 << 16, canonical=can) as t:
                t.set_mode(mode)
                t.attach_bloom(b)
                t.count_ascii(seq)
                t.count_ascii(seq)                                # the same windows again: admitted ones are in the cache now
                t.sync()
                kk, cc = gpu.decode_records(t.dump_records(), k, 4)
                got = dict((O.to_str(np.array([a], dtype=np.uint64), k), c) for a, c in zip(kk.tolist(), cc.tolist()))
                assert got == {key: 2 * c for key, c in golden.items()}, (mode, log2_sets)
                assert t.stats().mers_fed == 2 * len(O.extract(seq, k, can))
                t.attach_bloom(None)


def test_cache_of_admitted_kmers_and_the_all_ones_key(gpu, monkeypatch):
    """k = 32 without -C: the all-T 32-mer is the key 2^64 - 1, whose cache tag (key + 1) is the empty way's 0 (round-5 advisor
    finding: it was admitted on an empty way without the counter being asked, and counted although seen once).  A counter fed
    everything once plus one half twice; exactly one all-T 32-mer in the once-only half: the filtered table must hold exactly the k-mers
    the oracle's check() > 1 admits (bloom_counter2.hpp:109-142, count_main.cc:109-119), poly-T not among them -- and
    with poly-T fed twice to the counter, among them."""
    monkeypatch.setenv("JFGPU_BLOOM_CACHE", "1")
    monkeypatch.setenv("JFGPU_BLOOM_CACHE_LOG2", "6")
    rng = random.Random(3)
    k = 32
    twice = "".join(rng.choice("ACGT") for _ in range(6000))
    once = "".join(rng.choice("ACGT") for _ in range(3000)) + "N" + "T" * 32 + "N" + "".join(rng.choice("ACGT") for _ in range(3000))
    for polyt_twice in (False, True):
        fed = (twice + "N" + twice + "N" + once + ("N" + "T" * 32 if polyt_twice else "")).encode()
        n = 20000
        with gpu.Bloom(k, gpu.opt_m(0.001, n), gpu.opt_k(0.001), canonical=False, seed=11) as b:
            b.insert_ascii(fed)
            b.sync()
            seq = (twice + "N" + once).encode()
            kmers = O.extract(seq, k, False)[:, 0]
            # what the ORACLE's check() says of the counter's bytes, key by key (oracle/jf_oracle.c: jfo_bc_check)
            uniq = np.unique(kmers)
            data = b.read()
            h0 = O.matrix_times(b.matrix1, 64, 2 * k, uniq.reshape(-1, 1))
            h1 = O.matrix_times(b.matrix2, 64, 2 * k, uniq.reshape(-1, 1))
            L = O.lib()
            adm = np.array([L.jfo_bc_check(data.ctypes.data, b.m, b.nb_hashes, x, y) > 1 for x, y in zip(h0.tolist(), h1.tolist())])
            exp_keys = set(uniq[adm].tolist())
            assert len(exp_keys) >= 5000
            for mode in (1, 2):
                with gpu.Table(k, 1 << 16, canonical=False) as t:
                    t.set_mode(mode)
                    t.attach_bloom(b)
                    t.count_ascii(seq)
                    t.count_ascii(seq)
                    t.sync()
                    kk, cc = gpu.decode_records(t.dump_records(), k, 4)
                    got = dict(zip(kk.tolist(), cc.tolist()))
                    t.attach_bloom(None)
                assert set(got) == exp_keys, (polyt_twice, mode)
                assert ((1 << 64) - 1 in got) == polyt_twice, (polyt_twice, mode)      # poly-T: only when the counter saw it twice
                occ = dict(zip(*[a.tolist() for a in np.unique(kmers, return_counts=True)]))
                assert got == {key: 2 * occ[key] for key in exp_keys}


def test_bloom_against_oracle_on_random_input(gpu):
    """Own random matrices, larger input with lower-case / N resets, saturation at 2, load() round trip."""
    rng = random.Random(41)
    k = 25
    seq = "".join(rng.choice("ACGTacgtN") for _ in range(200000)).encode()
    seq = seq + seq[:60000]                                   # part of it twice
    n = 150000
    m, nh = gpu.opt_m(0.001, n), gpu.opt_k(0.001)
    assert (m, nh) == (14 * n, 10)
    with gpu.Bloom(k, m, nh, canonical=True, seed=5) as b:
        b.insert_ascii(seq)
        b.sync()
        got = b.read()
        kmers = O.extract(seq, k, True)
        h0 = O.matrix_times(b.matrix1, 64, 2 * k, kmers)
        h1 = O.matrix_times(b.matrix2, 64, 2 * k, kmers)
        data = np.zeros(b.nb_bytes, dtype=np.uint8)
        L = O.lib()
        for x, y in zip(h0.tolist(), h1.tolist()):
            L.jfo_bc_insert(data.ctypes.data, m, nh, x, y)
        assert (got == data).all()
        assert got.max() <= 242
        with gpu.Bloom(k, m, nh, canonical=True, matrix1=b.matrix1, matrix2=b.matrix2) as c:
            c.load(got)
            sample = kmers[::97, 0]
            assert (c.keys(sample) == b.keys(sample)).all()
            assert (c.keys(sample, insert=True) == b.keys(sample)).all()      # insert returns the previous minimum


def test_cli_bc_and_count_bc(gpu, tmp_path):
    """jellyfish-amd bc writes a bloomcounter file the REFERENCE loads (ref_jf count --bc), and
    jellyfish-amd count --bc on the reference's golden .bc gives the reference's filtered dump."""
    cli = os.environ.get("JFGPU_CLI")
    if not cli:
        subprocess.check_call(["make", "-s", "cli"], cwd=ROOT)
        cli = os.path.join(ROOT, "bin", "jellyfish-amd")
    case = MANIFEST["bloom"][0]
    inp = os.path.join(GOLD, case["input"])
    out = str(tmp_path / "f.jf")
    subprocess.check_call([cli, "count", "-m", "21", "-C", "-s", "64k", "--bc", os.path.join(GOLD, case["ref_bc"]), "-o", out, inp])
    got = sorted(subprocess.check_output([cli, "dump", "-c", out]).decode().splitlines())
    assert got == open(os.path.join(GOLD, case["name"] + ".filtered.dump")).read().splitlines()
    mine = str(tmp_path / "mine.bc")
    subprocess.check_call([cli, "bc", "-m", "21", "-C", "-s", "9000", "-o", mine, inp])
    header, body = read_bc(mine)
    assert header["format"] == "bloomcounter" and header["size"] == 126000 and header["nb_hashes"] == 10
    assert len(body) == 25200 and header["matrix1"]["r"] == 64 and len(header["matrix2"]["columns"]) == 42
    out2 = str(tmp_path / "f2.jf")
    subprocess.check_call([cli, "count", "-m", "21", "-C", "-s", "64k", "--bc", mine, "-o", out2, inp])
    a = sorted(subprocess.check_output([cli, "dump", "-c", out2]).decode().splitlines())
    if O.have_ref():
        out3 = str(tmp_path / "f3.jf")
        subprocess.check_call([O.REF_JF, "count", "-m", "21", "-C", "-s", "64k", "--bc", mine, "-o", out3, inp])
        b = sorted(subprocess.check_output([O.REF_JF, "dump", "-c", out3]).decode().splitlines())
        assert a == b
    # every k-mer that truly occurs at least twice survives any correct filter
    seq = O.parse_file(open(inp, "rb").read())
    keys, cnt = O.count(seq, 21, True)
    twice = {O.to_str(keys[i], 21) for i in range(len(keys)) if cnt[i] >= 2}
    assert twice <= {l.split()[0] for l in a}


def test_count_bc_single_pass_partition_equals_direct(gpu, monkeypatch):
    """count --bc through the single-pass partition (p1_scatter_granule_kernel<.., BLOOM=true>): same table
    as the direct kernel with the same filter attached, and k-mers seen once are filtered out."""
    monkeypatch.setenv("JFGPU_P1_SINGLE", "1")
    rng = random.Random(77)
    k = 16
    once = "".join(rng.choice("ACGT") for _ in range(300000))
    twice = "".join(rng.choice("ACGT") for _ in range(100000))
    seq = (once + "N" + twice + "N" + twice + "N" + "A" * 200).encode()       # includes a homopolymer run
    n = 500000
    with gpu.Bloom(k, gpu.opt_m(0.001, n), gpu.opt_k(0.001), canonical=True, seed=3) as b:
        b.insert_ascii(seq)
        b.sync()
        res = []
        for mode in (1, 2):
            with gpu.Table(k, 1 << 25, canonical=True) as t:
                t.set_mode(mode)
                t.attach_bloom(b)
                d = t.malloc(len(seq) + 64)
                t.h2d(d, np.frombuffer(seq, dtype=np.uint8))
                t.count_ascii_dev(d, len(seq))
                t.sync()
                st = t.stats()
                keys, cnts = gpu.decode_records(t.dump_records(), k, 4)
                res.append((st.distinct, st.total, dict(zip(keys.tolist(), cnts.tolist()))))
                t.attach_bloom(None)
                t.free(d)
        assert res[0] == res[1]
        kk, cc = O.count(seq, k, True)
        exact = dict(zip(kk[:, 0].tolist(), cc.tolist()))
        got = res[1][2]
        assert all(got.get(key, 0) == c for key, c in exact.items() if c >= 2)          # nothing seen twice is lost
        singles = [key for key, c in exact.items() if c == 1]
        assert sum(1 for key in singles if key in got) < 0.01 * len(singles)            # only false positives survive


@pytest.mark.parametrize("n_cells,two_level,share", [(14 * 150000, False, None), (14 * 30_000_000, True, None), (14 * 30_000_000, True, "4"),
                                                     (14 * 30_000_000, True, "single"), (14 * 30_000_000, True, "single4"),
                                                     (14 * 30_000_000, True, "ring10"), (14 * 30_000_000 + 3, True, "ring10"), (14 * 30_000_000 + 1, True, "ring5"), (5 * 3_400_000_000, True, "single")])
def test_partitioned_insert_equals_direct_and_oracle(gpu, monkeypatch, n_cells, two_level, share):
    """The partitioned insert (cell updates routed to 64 KiB segments, applied in LDS: kernels_bloom_part.hip.hpp) leaves
    the same bytes as one global compare-and-swap per cell and as the oracle's bloom_counter2 restatement; with more
    than 1024 segments the second partition level (P2) is on the path.  Several batches per flush, saturation at 2,
    a flush in the middle (check on encoded keys), inserts after it."""
    # share: P2 and the segment kernel go through the P1b buckets in groups that share one output buffer; single: the
    # single-pass P2 (fixed regions per segment, forced on: test-sized flushes would take the exact one)
    # ring10 / ring5 (round 6): P1b through rings of 256 bytes (p1_bloom_ring_kernel) forced onto a filter of 32 buckets -- every
    # round overflows its rings, so the straggler lists, their overflow into global compare-and-swaps and the final partial
    # units carry most of the updates; the 3.4 GB filter (206 of 256 buckets in use) takes the ring kernel by itself
    monkeypatch.setenv("JFGPU_P2_SINGLE", "2" if share and share.startswith("single") else "0")
    if share and share.startswith("ring"):
        monkeypatch.setenv("JFGPU_BLOOM_P1_RING", "2" if share == "ring10" else "3")
    if share and share[-1] == "4":
        monkeypatch.setenv("JFGPU_FLUSH_SHARE", "4")
    rng = random.Random(17)
    k, nh = 31, 10
    seq = ("".join(rng.choice("ACGT") for _ in range(90000)) + "N" + "".join(rng.choice("ACGTacgtN") for _ in range(30000))).encode()
    seq = seq + seq[:50000] + seq[:20000]                      # parts of it two and three times: cells saturate
    out = {}
    for mode in (1, 2):
        with gpu.Bloom(k, n_cells, nh, canonical=True, seed=9) as b:
            b.set_mode(mode)
            b.profile_enable(True)
            third = len(seq) // 3
            b.insert_ascii(seq[:third])
            b.insert_ascii(seq[third - (k - 1): 2 * third])
            kmers = O.extract(seq[:third], k, True)
            assert (b.keys(kmers[::501, 0]) >= 1).all()          # flushes what is pending
            b.insert_ascii(seq[2 * third - (k - 1):])
            assert b.sync() == len(O.extract(seq, k, True))
            out[mode] = b.read()
            used = [b.profile_get(i)[1] for i in range(4)]
            if mode == 2:
                assert used[1] > 0 and used[3] > 0 and (used[2] > 0) == two_level, used
            else:
                assert used[1] == 0 and used[0] > 0
            m1, m2 = b.matrix1, b.matrix2
    assert (out[1] == out[2]).all()
    if not two_level:
        kmers = O.extract(seq, k, True)
        h0, h1 = O.matrix_times(m1, 64, 2 * k, kmers), O.matrix_times(m2, 64, 2 * k, kmers)
        data = np.zeros(len(out[1]), dtype=np.uint8)
        L = O.lib()
        for x, y in zip(h0.tolist(), h1.tolist()):
            L.jfo_bc_insert(data.ctypes.data, n_cells, nh, x, y)
        assert (out[2] == data).all()


@pytest.mark.parametrize("k,single", [(31, "1"), (31, "0"), (21, "1"), (16, "1")])
def test_count_bc_through_every_p1_variant(gpu, monkeypatch, k, single):
    """count --bc with the filter evaluated as a per-lane mask in rounds (bloom_admit_mask): the single-pass P1 with 64-bit
    items (k = 31 here), the exact two-pass P1, the 32-bit single-pass P1 and the direct kernel admit exactly the k-mers
    the oracle's check() > 1 admits -- on a filter that holds more than the k-mers counted, so false positives exist."""
    monkeypatch.setenv("JFGPU_P1_SINGLE", single)
    rng = random.Random(k)
    seq = "".join(rng.choice("ACGT") for _ in range(250000)).encode()
    seq = seq + b"N" + seq[:80000] + b"N" + "".join(rng.choice("ACGTN") for _ in range(50000)).encode()
    n = 40000                                                  # undersized on purpose: fp rate far above 0.001
    m, nh = gpu.opt_m(0.01, n), gpu.opt_k(0.01)
    with gpu.Bloom(k, m, nh, canonical=True, seed=3) as b:
        b.insert_ascii(seq)
        b.sync()
        kmers = O.extract(seq, k, True)
        keys, cnt = O.count(seq, k, True)
        chk = b.keys(keys[:, 0])
        exp = {int(a): int(c) for a, c, ok in zip(keys[:, 0], cnt, chk == 2) if ok}
        assert len(exp) > int((cnt >= 2).sum())                # some singletons pass: the mask is not trivially right
        for mode in (1, 2):
            with gpu.Table(k, 1 << 25, canonical=True) as t:
                t.set_mode(mode)
                t.attach_bloom(b)
                t.count_ascii(seq)
                t.sync()
                kk, cc = gpu.decode_records(t.dump_records(), k, 4)
                assert dict(zip(kk.tolist(), cc.tolist())) == exp, (k, single, mode)
                assert t.stats().mers_fed == len(kmers)
                t.attach_bloom(None)


@pytest.mark.parametrize("k,mode", [(21, 1), (21, 2), (40, 1), (40, 2), (31, 2)])
def test_one_pass_bloom_filter(gpu, monkeypatch, k, mode):
    """count --bf-size (count_main.cc:121-131, bloom_filter.hpp:44-68): the first sighting of a k-mer only marks it in a
    Bloom filter of bits, later sightings are counted.  With the second copy of the input fed in a later call the
    semantics are exact up to false positives: every k-mer of the repeated part is counted once (twice if its first
    sighting was a false positive), k-mers seen once are absent but for false positives -- the bound of the reference's
    tests/bloom_filter.sh.  Direct kernel and single-pass partition (the two-pass P1 would ask the filter twice)."""
    monkeypatch.setenv("JFGPU_P1_SINGLE", "1")
    rng = random.Random(k * 7 + mode)
    twice = "".join(rng.choice("ACGT") for _ in range(120000)).encode()
    once = "".join(rng.choice("ACGT") for _ in range(60000)).encode()
    kt, _ = O.count(twice, k, True)
    ko, _ = O.count(once, k, True)
    rep = {tuple(r) for r in kt.tolist()}
    single = {tuple(r) for r in ko.tolist()} - rep
    n = len(rep) + len(single)
    with gpu.Bloom(k, gpu.opt_m(0.01, n), gpu.opt_k(0.01), canonical=True, one_pass_filter=True) as bf, \
            gpu.Table(k, 1 << 25, canonical=True) as t:
        assert bf.nb_bytes == (gpu.opt_m(0.01, n) + 7) // 8 and bf.nb_hashes == 7
        t.set_mode(mode)
        t.attach_bloom(bf)
        t.count_ascii(twice + b"N" + once)
        t.sync()                                               # first sightings are in the filter now
        t.count_ascii(twice)
        t.sync()
        kk, cc = gpu.decode_records(t.dump_records(), k, 4)
        got = {(tuple(r) if kk.ndim == 2 else (r,)): c for r, c in zip(kk.tolist(), cc.tolist())}
        t.attach_bloom(None)
    mult = {}
    kt_all = O.extract(twice, k, True)
    for r in kt_all.tolist():
        mult[tuple(r)] = mult.get(tuple(r), 0) + 1
    # every repeated k-mer is there: all its sightings of the second feed, plus those of the first feed that were false positives
    assert all(mult[key] <= got.get(key, 0) <= 2 * mult[key] for key in rep)
    extra = sum(got[key] - mult[key] for key in rep)
    fp_single = sum(1 for key in single if key in got)
    assert extra <= 0.03 * len(kt_all) and fp_single <= 0.03 * len(single)
    assert set(got) <= rep | single


@pytest.mark.parametrize("k,world,items", [(21, 2, "2"), (21, 4, "0"), (31, 2, "0")])
def test_sharded_count_with_a_bloom_counter_equals_the_single_table(gpu, monkeypatch, k, world, items):
    """`count --bc` over hash-prefix shards (count_main.cc:109-119 with --gpus; round-3 review, missing #1): every rank
    holds the whole read-only Bloom counter and asks it on the SENDING side -- in the routing kernels of the item path
    (k = 21: p1_ring_kernel<.., BLOOM, RouteListDirect>) and of the key path (partition_count / scatter_kernel<BLOOM>) --
    so what the filter does not admit never travels.  The shards together hold exactly what one table with the same
    counter attached holds."""
    monkeypatch.setenv("JFGPU_COMM_ITEMS", items)
    rng = random.Random(k * 3 + world)
    once = "".join(rng.choice("ACGT") for _ in range(120000))
    twice = "".join(rng.choice("ACGT") for _ in range(60000))
    steps = [[(twice[i * 15000:(i + 1) * 15000] + "N" + once[(4 * i + r) * 5000:(4 * i + r + 1) * 5000] + "N" + twice[i * 15000:(i + 1) * 15000]).encode()
              for r in range(world)] for i in range(4)]
    whole_seq = b"N".join(b"N".join(step) for step in steps)
    n = 400000
    with gpu.Bloom(k, gpu.opt_m(0.01, n), gpu.opt_k(0.01), canonical=True, seed=9) as b:
        b.insert_ascii(whole_seq)
        b.sync()
        with gpu.Table(k, 1 << 22, canonical=True) as single:
            single.attach_bloom(b)
            single.count_ascii(whole_seq); single.sync()
            kk, cc = gpu.decode_records(single.dump_records(), k, 4)
            exp = dict(zip(kk.tolist(), cc.tolist()))
            single.attach_bloom(None)
        keys, cnt = O.count(whole_seq, k, True)
        assert 1000 < len(exp) < len(keys), "the filter must admit some k-mers and refuse others"
        sb = world.bit_length() - 1
        shards = [gpu.Table(k, 1 << 22, canonical=True, shard_bits=sb, shard_id=r) for r in range(world)]
        comm = gpu.Comm(world, local=True)
        try:
            for t in shards:
                t.attach_bloom(b)
            bufs = []
            for step in steps:
                ptrs, ns = [], []
                for r, seq in enumerate(step):
                    d = shards[r].malloc(len(seq) + 64)
                    shards[r].h2d(d, np.frombuffer(seq, dtype=np.uint8))
                    bufs.append((shards[r], d)); ptrs.append(d); ns.append(len(seq))
                comm.local_step(shards, ptrs, ns)
            sent, received = comm.finish()
            assert sent == received == sum(exp.values())
            got = {}
            for t in shards:
                t.sync()
                kk, cc = gpu.decode_records(t.dump_records(), k, 4)
                part = dict(zip(kk.tolist(), cc.tolist()))
                assert not (set(part) & set(got))
                got.update(part)
            assert got == exp
            for t, d in bufs:
                t.free(d)
        finally:
            comm.close()
            for t in shards:
                t.attach_bloom(None)
                t.close()


@pytest.mark.parametrize("world", [2, 4])
def test_bloom_counters_of_the_ranks_merge_into_the_counter_of_the_whole_input(gpu, world):
    """`bc` over several GPUs (jfgpu_comm_bc_merge_local: all ranks in this process, device copies as the transport; the
    same code as the RCCL / ipc transports' jfgpu_comm_bc_merge): every rank inserts its part of the input into its own
    counter, the merge makes every counter the counter of the WHOLE input -- checked against the oracle's insert__
    (bloom_counter2.hpp:56-107) fed with everything in order: k-mers seen once on two ranks read 2, seen twice on one and
    once on another still 2 (saturation), and the k-mer tally is the sum."""
    rng = random.Random(91 + world)
    k = 25
    shared = "".join(rng.choice("ACGT") for _ in range(4000))            # on every rank: cells reach 2 by addition
    parts = []
    for r in range(world):
        own = "".join(rng.choice("ACGTN") for _ in range(30000))
        parts.append((own + "N" + shared + ("N" + shared if r == 1 else "")).encode())      # (rank 1 has it twice: 2 + 1 saturates)
    whole = b"N".join(parts)
    n = 60000
    m, nh = gpu.opt_m(0.001, n), gpu.opt_k(0.001)
    comm = gpu.Comm(world, local=True)
    blooms = [gpu.Bloom(k, m, nh, canonical=True, seed=5) for _ in range(world)]
    try:
        for b, p in zip(blooms, parts):
            b.insert_ascii(p)
        comm.bc_merge(blooms)
        kmers = O.extract(whole, k, True)
        h0 = O.matrix_times(blooms[0].matrix1, 64, 2 * k, kmers)
        h1 = O.matrix_times(blooms[0].matrix2, 64, 2 * k, kmers)
        data = np.zeros(blooms[0].nb_bytes, dtype=np.uint8)
        L = O.lib()
        for x, y in zip(h0.tolist(), h1.tolist()):
            L.jfo_bc_insert(data.ctypes.data, m, nh, x, y)
        for b in blooms:
            assert (b.read() == data).all()
            assert b.sync() == len(kmers)
        ks, _ = O.count(shared.encode(), k, True)
        assert (blooms[world - 1].keys(ks[:, 0]) == 2).all()
    finally:
        comm.close()
        for b in blooms:
            b.close()


def np_matrix_times(cols, r, c, keys):
    """pos = M * key over GF(2) for an array of one-word keys, vectorised: M's images of the key's bytes from the oracle's
    matrix_times on the unit vectors (rectangular_binary_matrix.hpp:155-164 through oracle/jf_oracle.c), xor-combined."""
    units = O.matrix_times(cols, r, c, np.array([1 << i for i in range(c)], dtype=np.uint64))
    keys = np.asarray(keys, dtype=np.uint64)
    out = np.zeros(len(keys), dtype=np.uint64)
    for b in range((c + 7) // 8):
        tbl = np.zeros(256, dtype=np.uint64)
        for v in range(1, 256):
            low = v & -v
            bit = 8 * b + low.bit_length() - 1
            tbl[v] = tbl[v ^ low] ^ (units[bit] if bit < c else np.uint64(0))
        out ^= tbl[((keys >> np.uint64(8 * b)) & np.uint64(255)).astype(np.intp)]
    return out


def test_partitioned_insert_at_config_3_geometry(gpu, monkeypatch):
    """The Bloom pass at BASELINE configs[2]'s size (m = 14e10 cells = 28 GB: 2^9 P1b buckets of 2^10 segments): this is
    where the cell updates' P2 goes through the ring kernel (p2_ring_kernel<BloomRingDirect>, the array's partial last
    bucket through the sort) -- the filters of the other tests are too small for it.  Checked against the ORACLE's
    restatement of bloom_counter2.hpp:56-107 (insert__: cell_i = (h0 % m + i (h1 % m)) % m, five base-3 cells a byte,
    saturating at 2) evaluated sparsely -- the cells the input touches, by numpy from the oracle's own matrix products:
    the whole 28 GB array read back must hold exactly those bytes with exactly those values and zero everywhere else,
    and check__ (:109-142) of every k-mer asked about must be the minimum over its cells."""
    if os.environ.get("JFGPU_LIB"):
        pytest.skip("a filter of 28 GB: not under the host emulation")
    monkeypatch.setenv("JFGPU_P2_SINGLE", "2")             # (the input is small: single-pass P2 even though its regions are mostly head-room)
    rng = random.Random(77)
    k = 31
    once = "".join(rng.choice("ACGT") for _ in range(1_500_000))
    twice = "".join(rng.choice("ACGT") for _ in range(500_000))
    seq = (once + "N" + twice + "N" + twice).encode()
    tail = b"ACGTN" * 2000
    m, nh = 14 * 10_000_000_000, 10
    with gpu.Bloom(k, m, nh, canonical=True, seed=5) as b:
        b.set_mode(2)
        b.reserve(8 << 30)
        b.profile_enable(True); b.profile_reset()
        for rep in range(3):                                  # several batches pending before the flush
            b.insert_ascii(seq if rep == 0 else tail)
        n = b.sync()
        assert b.profile_get(2)[1] >= 1 and b.profile_get(3)[1] >= 1, "the partitioned stages (P2, segments) must have run"
        assert b.ring_p2_launches() >= 1, "the cell updates' P2 must have gone through the ring kernel"
        # the oracle, sparse: every occurrence of every k-mer bumps its nh cells
        kmers = np.concatenate([O.extract(seq, k, True), O.extract(tail, k, True), O.extract(tail, k, True)])[:, 0]
        assert n == len(kmers)
        h0 = np_matrix_times(b.matrix1, 64, 2 * k, kmers) % np.uint64(m)
        h1 = np_matrix_times(b.matrix2, 64, 2 * k, kmers) % np.uint64(m)
        sub = slice(0, 2000)                                  # (the vectorised products against the oracle's own, on a sample)
        assert (h0[sub] == O.matrix_times(b.matrix1, 64, 2 * k, kmers[sub]) % np.uint64(m)).all()
        assert (h1[sub] == O.matrix_times(b.matrix2, 64, 2 * k, kmers[sub]) % np.uint64(m)).all()
        cells = np.concatenate([(h0 + np.uint64(i) * h1) % np.uint64(m) for i in range(nh)])
        ucell, hits = np.unique(cells, return_counts=True)
        digit = np.minimum(hits, 2).astype(np.uint64)
        byte_of, val = ucell // np.uint64(5), digit * (np.uint64(3) ** (ucell % np.uint64(5)))
        ubyte, first = np.unique(byte_of, return_index=True)
        expect = np.add.reduceat(val, first).astype(np.uint8)
        got = b.read()
        assert len(got) == (m + 4) // 5
        assert (got[ubyte.astype(np.intp)] == expect).all(), "a touched byte of the array differs from bloom_counter2's"
        assert int(np.count_nonzero(got)) == len(ubyte), "cells were bumped that no k-mer of the input owns"
        # check__: the minimum over a k-mer's cells, for a sample of the k-mers inserted once and of those inserted twice
        keys1, _ = O.count(seq[:400_000], k, True)
        keys2, _ = O.count(twice[:200_000].encode(), k, True)
        keys = np.concatenate([keys1, keys2])[:, 0]; n1 = len(keys1)
        q0 = np_matrix_times(b.matrix1, 64, 2 * k, keys) % np.uint64(m)
        q1 = np_matrix_times(b.matrix2, 64, 2 * k, keys) % np.uint64(m)
        want = np.full(len(keys), 2, dtype=np.uint64)
        for i in range(nh):
            c = (q0 + np.uint64(i) * q1) % np.uint64(m)
            want = np.minimum(want, digit[np.searchsorted(ucell, c)])
        ans = b.keys(keys)
        assert (ans == want).all()
        assert (ans[n1:] == 2).all() and int((ans[:n1] == 1).sum()) > 0.99 * n1
