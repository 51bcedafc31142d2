"""BASELINE config 5 on the GPU (-m gpu): two-word keys, 33 <= k <= 64, 128-bit slots claimed with
64-bit atomics.  Bit-exact {k-mer -> count} against the multi-word oracle (the reference's
ceil(k/32)-word mer_dna, mer_dna.hpp:143-170; tests/large_key.sh is the reference's own check of
this path), dump in (pos, key) order, lookups, hash_counter::add, count-field overflow."""
import os
import random

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


def rnd_seq(rng, n, alphabet="ACGT"):
    return "".join(rng.choice(alphabet) for _ in range(n)).encode()


def oracle_map(seq, k, canonical):
    keys, cnt = O.count(seq, k, canonical)
    return {tuple(r): c for r, c in zip(keys.tolist(), cnt.tolist())}


def table_map(capi, t):
    recs = t.dump_records(chunk_records=1 << 16)
    keys, cnts = capi.decode_records(recs, t.k, t.info.out_counter_len)
    assert len({tuple(r) for r in keys.tolist()}) == len(keys)
    if len(keys) > 1:                                          # (pos, key) order under the table's matrix
        sub = keys[:5000]
        pos = O.matrix_times(t.matrix(), t.info.lsize, 2 * t.k, sub)
        pk = [(p, r[1], r[0]) for p, r in zip(pos.tolist(), sub.tolist())]     # key compared from the top word
        assert pk == sorted(pk)
    return {tuple(r): c for r, c in zip(keys.tolist(), cnts.tolist())}


@pytest.mark.parametrize("k,canonical,n,alphabet", [
    (33, True, 40000, "ACGT"), (33, False, 20000, "ACGTN"), (40, True, 30000, "ACGTacgtNRY"),
    (48, True, 30000, "ACGT"), (63, True, 40000, "ACGT"), (63, False, 30000, "ACGTN"),
    (64, True, 30000, "ACGT"), (64, False, 4096 * 2 + 70, "ACGT"),
    (50, True, 60000, "AT"), (63, True, 50000, "A"), (40, True, 39, "ACGT"), (40, True, 40, "ACGT"),
])
def test_wide_count_matches_oracle(gpu, k, canonical, n, alphabet):
    rng = random.Random(k * 31 + n)
    seq = rnd_seq(rng, n, alphabet)
    exp = oracle_map(seq, k, canonical)
    with gpu.Table(k, 1 << 16, canonical=canonical) as t:
        assert t.info.slot_bytes == 16 and t.key_words == 2
        t.count_ascii(seq)
        t.sync()
        assert table_map(gpu, t) == exp
        st = t.stats()
        assert (st.distinct, st.total, st.mers_fed) == (len(exp), sum(exp.values()), sum(exp.values()))
        assert st.unique == sum(1 for v in exp.values() if v == 1)
        assert st.max_count == (max(exp.values()) if exp else 0)
        if exp:
            keys = np.array(list(exp.keys()), dtype=np.uint64)
            vals, found = t.lookup(keys)
            assert found.all() and vals.tolist() == list(exp.values())
            absent = np.array([[rng.getrandbits(64), rng.getrandbits(2 * k - 64)] for _ in range(500)], dtype=np.uint64)
            absent = np.array([r for r in absent.tolist() if tuple(r) not in exp], dtype=np.uint64)
            vals, found = t.lookup(absent)
            assert not found.any() and not vals.any()
        base, inc, h = t.histo(1, 100000, 1)
        ref = {}
        for v in exp.values():
            ref[v] = ref.get(v, 0) + 1
        assert {base + i * inc: int(c) for i, c in enumerate(h) if c} == ref


def test_wide_content_digest(gpu):
    rng = random.Random(11)
    seq = "".join(rng.choice("ACGTN") for _ in range(60000)).encode()
    for k in (33, 63):
        with gpu.Table(k, 1 << 16) as t:
            t.count_ascii(seq)
            t.sync()
            keys, cnts = gpu.decode_records(t.dump_records(), k, t.info.out_counter_len)
            assert t.digest() == gpu.digest_of(keys, cnts)


def test_wide_add_keys_and_overflow(gpu):
    k = 63
    rng = random.Random(3)
    with gpu.Table(k, 1 << 16, canonical=False) as t:
        keys = np.array([[rng.getrandbits(64), rng.getrandbits(62)] for _ in range(3000)], dtype=np.uint64)
        is_new = t.add_keys(keys, val=3, want_new=True)
        assert is_new.all()
        assert not t.add_keys(keys[:100], val=2, want_new=True).any()
        vals, found = t.lookup(keys)
        assert found.all() and vals[:100].tolist() == [5] * 100 and vals[100:].tolist() == [3] * 2900
        # same low word, different high word: distinct keys (exercises the lo/hi split of the tag)
        twins = np.array([[keys[0, 0], (int(keys[0, 1]) ^ 1)], [keys[0, 0] ^ np.uint64(1 << 40), keys[0, 1]]], dtype=np.uint64)
        vals, found = t.lookup(twins)
        assert not found.any()
        big = 2 ** 50 + 7                                      # beyond the in-slot count field: side table
        assert t.info.val_len < 50
        t.add_keys(twins[:1], val=big)
        t.add_keys(np.repeat(twins[1:2], 5000, axis=0), val=1)
        vals, found = t.lookup(twins)
        assert vals.tolist() == [big, 5000]
        st = t.stats()
        assert st.distinct == 3002 and st.max_count == big
        recs = t.dump_records()
        kk, cc = gpu.decode_records(recs, k, 4)
        got = {tuple(r): c for r, c in zip(kk.tolist(), cc.tolist())}
        assert got[tuple(twins[0].tolist())] == 2 ** 32 - 1 and got[tuple(twins[1].tolist())] == 5000


def test_wide_hash_full_and_lower_upper(gpu):
    rng = random.Random(5)
    k = 40
    seq = rnd_seq(rng, 150000, "ACGT")
    with gpu.Table(k, 1 << 13) as t:                           # engine minimum 2^13 slots < 150k distinct
        if t.info.size < 150000:
            t.set_growth(False)                                # --disk / do_size_doubling(false)
            t.count_ascii(seq)
            with pytest.raises(gpu.JfgpuError) as e:
                t.sync()
            assert e.value.code == gpu.E_FULL
    seq = rnd_seq(rng, 30000, "AC")
    exp = oracle_map(seq, 34, True)
    with gpu.Table(34, 1 << 16) as t:
        t.count_ascii(seq)
        t.sync()
        recs = t.dump_records(2, 3)
        kk, cc = gpu.decode_records(recs, 34, 4)
        assert {tuple(r): c for r, c in zip(kk.tolist(), cc.tolist())} == {a: b for a, b in exp.items() if 2 <= b <= 3}


@pytest.mark.parametrize("k,canonical,n", [(40, True, 150000), (33, False, 120000), (48, True, 100000)])
def test_wide_table_grows_like_the_reference(gpu, k, canonical, n):
    """The size is a hint for two-word keys too (tests/large_key.sh: `-s 2k -m 100` must equal `-s 2M`): fed in
    several calls with look-ups in between, the table doubles as often as needed, counts stay exact (large values
    in the overflow side table included) and the dump is ordered under the final matrix."""
    rng = random.Random(k + n)
    seq = rnd_seq(rng, n, "ACGTN")
    exp = oracle_map(seq, k, canonical)
    with gpu.Table(k, 1 << 13, canonical=canonical) as t:
        first = t.info.lsize
        third = len(seq) // 3
        t.count_ascii(seq[:third])
        some = np.array(list(exp.keys())[:500], dtype=np.uint64)
        t.lookup(some)
        t.add_keys(some[:5], val=2 ** 50)
        t.count_ascii(seq[third - (k - 1):2 * third])
        t.count_ascii(seq[2 * third - (k - 1):])
        t.sync()
        assert (t.info.lsize > first or (1 << first) * 0.8 >= len(exp)) and (1 << t.info.lsize) >= len(exp)   # k = 64 starts at 2^31
        got = table_map(gpu, t)
        want = dict(exp)
        for key in some[:5].tolist():
            want[tuple(key)] += 2 ** 50
        capped = {a: min(b, 2 ** 32 - 1) for a, b in want.items()}          # the dump saturates at 4 bytes
        assert got == capped
        vals, found = t.lookup(some)
        assert found.all() and vals.tolist() == [want[tuple(x)] for x in some.tolist()]
        st = t.stats()
        assert (st.distinct, st.total) == (len(exp), sum(want.values()))


@pytest.mark.parametrize("k,size", [(33, 1 << 16), (40, 1 << 20), (63, 1 << 16)])
def test_wide_partitioned_path_equals_direct_and_oracle(gpu, k, size):
    """Two-word keys through the partitioned insert (kernels_wide_part.hip.hpp: 128-bit items, 128 KiB tiles in LDS, the
    two-word claim on LDS words) against the oracle and against the global-atomic path, in the same table format:
    several batches per flush, duplicates far apart, low-complexity stretches that overflow a bucket region, a flush in
    the middle, direct inserts on top of tiles written by the tile kernel.  k = 63 starts at 2^29 slots, i.e. with the
    second partition level; the small tables are single-level."""
    rng = random.Random(k)
    seq = rnd_seq(rng, 70000, "ACGT") + b"N" + rnd_seq(rng, 20000, "ACGTacgtN") + b"A" * 3000 + rnd_seq(rng, 5000, "AC")
    seq = seq + seq[:30000]
    exp = oracle_map(seq, k, True)
    dumps = {}
    for mode in (1, 2):
        with gpu.Table(k, size, canonical=True) as t:
            t.set_mode(mode)
            t.profile_enable(True)
            half = len(seq) // 2
            t.count_ascii(seq[:half])
            sample = np.array(list(exp.keys())[:200], dtype=np.uint64)
            t.lookup(sample)                                   # flush + read in the middle
            t.count_ascii(seq[half - (k - 1):])
            t.sync()
            used = [t.profile_get(i)[1] for i in range(8)]
            if mode == 2:
                assert used[4] > 0 and (used[6] > 0 or used[7] > 0) and used[0] == 0, used
            else:
                assert used[0] > 0 and used[4] == 0, used
            assert table_map(gpu, t) == exp
            st = t.stats()
            assert (st.distinct, st.total, st.mers_fed) == (len(exp), sum(exp.values()), sum(exp.values()))
            t.add_keys(sample[:50], val=3)                     # global two-word claim on top of LDS-written tiles
            vals, found = t.lookup(sample[:50])
            assert found.all() and vals.tolist() == [exp[tuple(r)] + 3 for r in sample[:50].tolist()]
            dumps[mode] = t.digest()
    assert dumps[1] == dumps[2]


def test_wide_partitioned_grows_and_filters(gpu):
    """A tiny size hint with two-word keys in partitioned mode: the table doubles itself between flushes; and count --bc
    (Bloom filter attached) admits the same k-mers through both insert strategies."""
    rng = random.Random(3)
    k = 40
    seq = rnd_seq(rng, 150000, "ACGT")
    seq = seq + seq[:50000]
    exp = oracle_map(seq, k, True)
    with gpu.Table(k, 1 << 13) as t:
        t.set_mode(2)
        first = t.info.lsize
        for a in range(0, len(seq), 40000):
            t.count_ascii(seq[max(0, a - (k - 1)): a + 40000])
        t.sync()
        assert t.info.lsize > first
        assert table_map(gpu, t) == exp
    n = len(exp)
    with gpu.Bloom(k, gpu.opt_m(0.001, n), gpu.opt_k(0.001)) as b:
        b.insert_ascii(seq)
        b.sync()
        got = {}
        for mode in (1, 2):
            with gpu.Table(k, 1 << 18) as t:
                t.set_mode(mode)
                t.attach_bloom(b)
                t.count_ascii(seq)
                t.sync()
                got[mode] = table_map(gpu, t)
                t.attach_bloom(None)
        assert got[1] == got[2]
        twice = {key: c for key, c in exp.items() if c >= 2}
        assert all(got[2].get(key) == c for key, c in twice.items())          # no false negatives


@pytest.mark.parametrize("k,canonical,n,alphabet", [
    (100, False, 20000, "ACGT"), (65, True, 15000, "ACGT"), (96, False, 9000, "ACGTacgt"), (97, True, 30000, "AC"),
    (120, True, 40000, "ACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTN"), (80, True, 79, "ACGT"), (80, True, 80, "ACGT"),
])
def test_keys_of_three_and_four_words(gpu, k, canonical, n, alphabet):
    """65 <= k <= 128 (kernels_nword.hip.hpp; the reference's tests/large_key.sh range): 256-bit keys in four-word slots
    claimed word by word, a size hint far too small (the table doubles itself), counts, (pos, key) order of the dump
    under the final matrix, look-ups with keys of ceil(2k/64) words, the content digest, hash_counter::add."""
    rng = random.Random(k + n)
    seq = rnd_seq(rng, n, alphabet)
    keys, cnt = O.count(seq, k, canonical)
    exp = {tuple(r): c for r, c in zip(keys.tolist(), cnt.tolist())}
    with gpu.Table(k, 1 << 12, canonical=canonical) as t:
        assert t.info.slot_bytes == 32 and t.key_words == (2 * k + 63) // 64
        t.count_ascii(seq[: n // 2])
        t.count_ascii(seq[max(0, n // 2 - (k - 1)):])
        t.sync()
        st = t.stats()
        assert (st.distinct, st.total, st.mers_fed) == (len(exp), sum(exp.values()), sum(exp.values()))
        kk, cc = gpu.decode_records(t.dump_records(chunk_records=1 << 16), k, t.info.out_counter_len)
        assert {tuple(r): c for r, c in zip(kk.tolist(), cc.tolist())} == exp
        assert t.digest() == gpu.digest_of(keys, cnt)
        if len(kk) > 1:
            pos = O.matrix_times(t.matrix(), t.info.lsize, 2 * k, kk[:4000])
            pk = [(p,) + tuple(reversed(r)) for p, r in zip(pos.tolist(), kk[:4000].tolist())]
            assert pk == sorted(pk)
            vals, found = t.lookup(kk[:300])
            assert found.all() and vals.tolist() == cc[:300].tolist()
            absent = kk[:50].copy(); absent[:, 0] ^= np.uint64(0x5555)
            absent = np.array([r for r in absent.tolist() if tuple(r) not in exp], dtype=np.uint64).reshape(-1, kk.shape[1])
            vals, found = t.lookup(absent)
            assert not found.any()
            new = t.add_keys(np.concatenate([kk[:20], absent[:5]]), val=2 ** 40 + 1, want_new=True)
            assert new.tolist() == [0] * 20 + [1] * len(absent[:5])
            vals, found = t.lookup(kk[:20])
            assert vals.tolist() == [int(c) + 2 ** 40 + 1 for c in cc[:20]]


@pytest.mark.parametrize("k", [35, 21, 100])
def test_dump_of_saturated_count_fields_over_all_ones_tags(gpu, k):
    """A slot whose count field is all ones (hash_counter::add(m, UINT64_MAX), what unit_tests/test_hash_counter.cc does)
    over a tag whose stored bits are all ones is a word of all ones -- it must still come out of the sorted dump (the dump's
    'empty' marker once was all ones: such records were counted, then skipped, and the output kept uninitialised bytes)."""
    rng = random.Random(k)
    kw = (2 * k + 63) // 64
    top = 2 * k - 64 * (kw - 1)
    keys = np.array([[rng.getrandbits(64) for _ in range(kw - 1)] + [rng.getrandbits(top)] for _ in range(600)], dtype=np.uint64)
    with gpu.Table(k, 1 << 16, canonical=False, out_counter_len=8) as t:
        t.add_keys(keys[:, 0].copy() if kw == 1 else keys, val=2 ** 64 - 1)
        t.sync()
        recs = t.dump_records()
        kk, cc = gpu.decode_records(recs, k, 8)
        assert sorted(map(tuple, np.asarray(kk).reshape(len(cc), -1).tolist())) == sorted(map(tuple, keys.tolist()))
        assert set(cc.tolist()) == {2 ** 64 - 1}


def test_flush_in_groups_sharing_one_p2_buffer(gpu, monkeypatch):
    """When the arena cannot hold a whole flush's P2 output the P1 buckets go through P2 and the tile insert in groups that
    reuse one small buffer (what lets 10 Gbp of 63-mers flush once).  Forced here with 4 groups: two-word and one-word
    keys, two flushes (the second one loads dirty tiles), per-stage timers on; the table equals the direct path's."""
    rng = random.Random(12)
    for k, size in ((40, 1 << 26), (21, 1 << 27)):
        seq = rnd_seq(rng, 200000, "ACGT") + b"N" + rnd_seq(rng, 50000, "AC")
        digests = {}
        for share in ("4", "4s", "0"):                       # 4s: the single-pass P2 (fixed regions), also in 4 groups
            monkeypatch.setenv("JFGPU_FLUSH_SHARE", share[0])
            monkeypatch.setenv("JFGPU_P2_SINGLE", "2" if share == "4s" else "0")
            with gpu.Table(k, size, canonical=True) as t:
                t.set_mode(2 if share != "0" else 1)
                t.profile_enable(True)
                half = len(seq) // 2
                t.count_ascii(seq[:half]); t.sync()
                t.count_ascii(seq[half - (k - 1):]); t.sync()
                if share != "0":
                    assert t.profile_get(5)[1] >= 4 and t.profile_get(6)[1] >= 4        # P2 and T ran per group
                st = t.stats()
                digests[share] = (t.digest(), st.total, st.distinct)
        assert digests["4"] == digests["0"] == digests["4s"]


@pytest.mark.parametrize("k,world,max_msg", [(40, 2, None), (48, 4, "997"), (33, 1, None), (63, 2, None)])
def test_sharded_two_word_keys_equal_single_table(gpu, monkeypatch, k, world, max_msg):
    """Hash-prefix shards of a table of two-word keys (round 4; round 3 refused k > 32 with shard_bits): every rank routes
    the 128-bit k-mers of its input by owner (partition_count / scatter_wide_kernel), the messages carry two words per
    k-mer, receivers insert with the two-word claim.  The shards' dumps concatenated in rank order are byte-identical to the
    dump of one table of the global size under the same matrix, and nothing is lost or duplicated."""
    if max_msg:
        monkeypatch.setenv("JFGPU_COMM_MAX_MSG", max_msg)
    if k == 63 and os.environ.get("JFGPU_LIB"):
        pytest.skip("k = 63 needs 2^29 slots (8 GB): not under the host emulation")
    rng = random.Random(k * 7 + world)
    inputs = [[rnd_seq(rng, rng.choice([0, 70, 20000, 40000]), "ACGTN") for _ in range(world)] for _step in range(3)]
    inputs[1][0] = b""
    inputs[0][world - 1] = rnd_seq(rng, 30000, "ACGT")
    whole_seq = b"N".join(b"N".join(step) for step in inputs)
    keys, cnt = O.count(whole_seq, k, True)
    assert len(keys) > 1000
    with gpu.Table(k, 1 << 18) as single:
        single.set_growth(False)
        single.count_ascii(whole_seq); single.sync()
        whole = single.dump_records()
        cols = single.matrix()
        lsize_g = single.info.lsize                           # (the engine raises the size to the slot format's minimum: 2^29 at k = 63)
    sb = world.bit_length() - 1
    shards = [gpu.Table(k, 1 << lsize_g, shard_bits=sb, shard_id=r, matrix_columns=cols) for r in range(world)]
    comm = gpu.Comm(world, local=True)
    try:
        assert all(t.info.lsize == lsize_g for t in shards)
        bufs = []
        for step in inputs:
            ptrs, ns = [], []
            for r, seq in enumerate(step):
                d = shards[r].malloc(len(seq) + 64)
                if seq:
                    shards[r].h2d(d, np.frombuffer(seq, dtype=np.uint8))
                bufs.append((shards[r], d)); ptrs.append(d); ns.append(len(seq))
            comm.local_step(shards, ptrs, ns)
        sent, received = comm.finish()
        assert sent == received == int(cnt.sum())
        for t in shards:
            t.sync()
        parts = [t.dump_records() for t in shards]
        assert (np.concatenate(parts) == whole).all()
        assert sum(t.stats().total for t in shards) == int(cnt.sum())
        assert sum(t.stats().distinct for t in shards) == len(keys)
        # a k-mer sent to a shard that does not own it is refused, never silently counted
        foreign = np.array([[int(keys[0][0]), int(keys[0][1])]], dtype=np.uint64)
        refused = 0
        for t in shards:
            try:
                t.add_keys(foreign)
                t.sync()
            except gpu.JfgpuError as e:
                assert "does not own" in e.msg
                refused += 1
        assert refused == world - 1
        for t, d in bufs:
            t.free(d)
    finally:
        comm.close()
        for t in shards:
            t.close()


@pytest.mark.parametrize("k,world", [(40, 2), (48, 4), (33, 1)])
def test_two_word_shards_grow_together(gpu, k, world):
    """hash_counter::double_size (hash_counter.hpp:200-238) for sharded tables of two-word keys: shards created far too
    small double together (abi_comm.inl: comm_grow -- reshard_wide_kernel, pairs of (two key words, count) through the
    key path's exchange, add_pairs_wide_kernel).  What the shards hold afterwards is what one table that grew on its own
    holds: the same k-mers with the same counts, every k-mer on the shard its position names, nothing lost in transit."""
    rng = random.Random(k * 11 + world)
    steps = [[rnd_seq(rng, rng.choice([20000, 50000, 80000]), "ACGT") + b"N" + rnd_seq(rng, 300, "ACGTN") for _ in range(world)] for _step in range(4)]
    whole_seq = b"N".join(b"N".join(step) for step in steps)
    keys, cnt = O.count(whole_seq, k, True)
    exp = oracle_map(whole_seq, k, True)
    sb = world.bit_length() - 1
    shards = [gpu.Table(k, 1 << 14, shard_bits=sb, shard_id=r) for r in range(world)]
    comm = gpu.Comm(world, local=True)
    try:
        lsize0 = shards[0].info.lsize
        bufs = []
        for step in steps:
            ptrs, ns = [], []
            for r, seq in enumerate(step):
                d = shards[r].malloc(len(seq) + 64)
                shards[r].h2d(d, np.frombuffer(seq, dtype=np.uint8))
                bufs.append((shards[r], d)); ptrs.append(d); ns.append(len(seq))
            comm.local_step(shards, ptrs, ns)
        sent, received = comm.finish()
        assert sent == received == int(cnt.sum())
        got = {}
        for r, t in enumerate(shards):
            t.sync()
            assert t.info.lsize > lsize0 + 2 and t.info.lsize == shards[0].info.lsize, "the shards must have doubled, and together"
            part = table_map(gpu, t)                          # (checks the (pos, key) order under the table's own matrix)
            sub = np.array(list(part.keys())[:4000], dtype=np.uint64).reshape(-1, 2)
            if sb and len(sub):
                pos = O.matrix_times(t.matrix(), t.info.lsize, 2 * k, sub)
                assert ((pos >> np.uint64(t.info.lsize - sb)) == r).all()
            assert not (set(part) & set(got))
            got.update(part)
        assert got == exp
        assert len({tuple(t.matrix().tolist()) for t in shards}) == 1
        assert sum(t.stats().total for t in shards) == int(cnt.sum())
        for t, d in bufs:
            t.free(d)
    finally:
        comm.close()
        for t in shards:
            t.close()
