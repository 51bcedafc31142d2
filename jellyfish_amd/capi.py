"""ctypes binding of the C ABI in include/jfgpu.h (jellyfish_amd/lib/libjfgpu.so).

This is plumbing for tests/ and bench.py: one Python method per C entry point, no
logic of its own and NO fallback -- if the HIP library is missing or there is no
GPU, calls raise.  The reference-shaped host API (mer_dna / hash_counter /
file_header / dumpers) is the C++ facade under jellyfish_amd/include/.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("JFGPU_LIB") or os.path.join(_HERE, "lib", "libjfgpu.so")   # JFGPU_LIB: experimental builds (tools/)

OK, E_INVALID, E_NO_DEVICE, E_ALLOC, E_FULL, E_HIP, E_UNSUPPORTED, E_FORMAT = range(8)
PARSE_FASTA, PARSE_FASTQ, PARSE_CONTINUE = 1, 2, 4


class JfgpuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"jfgpu error {code}: {msg}")
        self.code = code
        self.msg = msg


class Params(C.Structure):
    _fields_ = [
        ("k", C.c_uint32), ("canonical", C.c_uint32), ("size", C.c_uint64), ("device", C.c_int32),
        ("shard_bits", C.c_uint32), ("shard_id", C.c_uint32), ("matrix_seed", C.c_uint64),
        ("matrix_columns", C.POINTER(C.c_uint64)), ("out_counter_len", C.c_uint32), ("matrix_kind", C.c_uint32),
    ]


class Info(C.Structure):
    _fields_ = [
        ("k", C.c_uint32), ("key_len", C.c_uint32), ("canonical", C.c_uint32), ("lsize", C.c_uint32),
        ("size", C.c_uint64), ("local_size", C.c_uint64), ("shard_bits", C.c_uint32), ("shard_id", C.c_uint32),
        ("val_len", C.c_uint32), ("slot_bytes", C.c_uint32), ("tile_slots", C.c_uint32),
        ("matrix_identity", C.c_uint32), ("out_counter_len", C.c_uint32), ("max_reprobe", C.c_uint32),
        ("table_bytes", C.c_uint64),
    ]


class BloomParams(C.Structure):
    _fields_ = [("k", C.c_uint32), ("canonical", C.c_uint32), ("m", C.c_uint64), ("nb_hashes", C.c_uint32),
                ("device", C.c_int32), ("seed", C.c_uint64), ("matrix1", C.POINTER(C.c_uint64)),
                ("matrix2", C.POINTER(C.c_uint64))]


class Stats(C.Structure):
    _fields_ = [("unique", C.c_uint64), ("distinct", C.c_uint64), ("total", C.c_uint64),
                ("max_count", C.c_uint64), ("occupied", C.c_uint64), ("mers_fed", C.c_uint64)]


# name -> (restype, argtypes); also the list tests check against include/jfgpu.h
_P = C.c_void_p
SIGNATURES = {
    "jfgpu_last_error": (C.c_char_p, []),
    "jfgpu_abi_version": (C.c_int, []),
    "jfgpu_device_count": (C.c_int, []),
    "jfgpu_create": (C.c_int, [C.POINTER(Params), C.POINTER(_P)]),
    "jfgpu_destroy": (None, [_P]),
    "jfgpu_get_info": (C.c_int, [_P, C.POINTER(Info)]),
    "jfgpu_get_matrix": (C.c_int, [_P, _P]),
    "jfgpu_clear": (C.c_int, [_P]),
    "jfgpu_sync": (C.c_int, [_P]),
    "jfgpu_wait": (C.c_int, [_P]),
    "jfgpu_count_ascii_dev": (C.c_int, [_P, _P, C.c_size_t]),
    "jfgpu_count_ascii": (C.c_int, [_P, _P, C.c_size_t]),
    "jfgpu_add_keys_dev": (C.c_int, [_P, _P, C.c_size_t, C.c_uint64, _P]),
    "jfgpu_add_keys": (C.c_int, [_P, _P, C.c_size_t, C.c_uint64, _P]),
    "jfgpu_add_key_vals": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "jfgpu_lookup_dev": (C.c_int, [_P, _P, C.c_size_t, _P, _P]),
    "jfgpu_lookup": (C.c_int, [_P, _P, C.c_size_t, _P, _P]),
    "jfgpu_partition_ascii_dev": (C.c_int, [_P, _P, C.c_size_t, _P, C.c_size_t, _P]),
    "jfgpu_comm_unique_id": (C.c_int, [_P]),
    "jfgpu_comm_create": (C.c_int, [C.c_int, C.c_int, _P, C.c_int, C.POINTER(_P)]),
    "jfgpu_comm_create_local": (C.c_int, [C.c_int, C.c_int, C.POINTER(_P)]),
    "jfgpu_comm_destroy": (None, [_P]),
    "jfgpu_comm_count_ascii_dev": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "jfgpu_comm_local_step": (C.c_int, [_P, _P, _P, _P]),
    "jfgpu_comm_finish": (C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "jfgpu_comm_allreduce_u64": (C.c_int, [_P, _P, C.c_int, C.c_int]),
    "jfgpu_comm_allgather_u64": (C.c_int, [_P, C.c_uint64, _P]),
    "jfgpu_comm_world": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "jfgpu_comm_exchange_times": (C.c_int, [_P, _P, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    "jfgpu_profile_spans": (C.c_int, [_P, _P, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    "jfgpu_stats_compute": (C.c_int, [_P, C.c_uint64, C.c_uint64, C.POINTER(Stats)]),
    "jfgpu_digest": (C.c_int, [_P, C.c_uint64, C.c_uint64, _P]),
    "jfgpu_histo": (C.c_int, [_P, C.c_uint64, C.c_uint64, C.c_uint64, _P, C.c_uint64]),
    "jfgpu_dump_begin": (C.c_int, [_P, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]),
    "jfgpu_dump_next": (C.c_int, [_P, _P, C.c_uint64, C.POINTER(C.c_uint64)]),
    "jfgpu_dump_end": (C.c_int, [_P]),
    "jfgpu_bc_opt_m": (C.c_uint64, [C.c_double, C.c_uint64]),
    "jfgpu_bc_opt_k": (C.c_uint32, [C.c_double]),
    "jfgpu_bc_create": (C.c_int, [C.POINTER(BloomParams), C.POINTER(_P)]),
    "jfgpu_bf_create": (C.c_int, [C.POINTER(BloomParams), C.POINTER(_P)]),
    "jfgpu_bc_destroy": (None, [_P]),
    "jfgpu_bc_insert_ascii_dev": (C.c_int, [_P, _P, C.c_size_t]),
    "jfgpu_bc_insert_ascii": (C.c_int, [_P, _P, C.c_size_t]),
    "jfgpu_bc_sync": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "jfgpu_bc_clear": (C.c_int, [_P]),
    "jfgpu_bc_get_info": (C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), _P, _P]),
    "jfgpu_bc_read": (C.c_int, [_P, _P]),
    "jfgpu_bc_load": (C.c_int, [_P, _P]),
    "jfgpu_bc_keys": (C.c_int, [_P, _P, C.c_size_t, _P, C.c_int]),
    "jfgpu_bc_set_mode": (C.c_int, [_P, C.c_int]),
    "jfgpu_bc_reserve": (C.c_int, [_P, C.c_uint64]),
    "jfgpu_bc_profile_enable": (C.c_int, [_P, C.c_int]),
    "jfgpu_bc_profile_get": (C.c_int, [_P, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "jfgpu_bc_profile_reset": (C.c_int, [_P]),
    "jfgpu_attach_bloom": (C.c_int, [_P, _P]),
    "jfgpu_comm_bc_merge": (C.c_int, [_P, _P]),
    "jfgpu_comm_bc_merge_local": (C.c_int, [_P, C.POINTER(_P)]),
    "jfgpu_parser_create": (C.c_int, [C.c_int, C.c_uint32, C.POINTER(_P)]),
    "jfgpu_parser_destroy": (None, [_P]),
    "jfgpu_parser_parse_dev": (C.c_int, [_P, _P, C.c_size_t, C.c_uint, C.POINTER(_P), C.POINTER(C.c_size_t), C.POINTER(C.c_uint64)]),
    "jfgpu_parser_parse": (C.c_int, [_P, _P, C.c_size_t, C.c_uint, C.POINTER(_P), C.POINTER(C.c_size_t), C.POINTER(C.c_uint64)]),
    "jfgpu_parser_upload": (C.c_int, [_P, C.c_int, _P, C.c_size_t]),
    "jfgpu_parser_upload_wait": (C.c_int, [_P, C.c_int]),
    "jfgpu_parser_parse_uploaded": (C.c_int, [_P, C.c_int, C.c_uint, C.POINTER(_P), C.POINTER(C.c_size_t), C.POINTER(C.c_uint64)]),
    "jfgpu_parser_host_buffer": (C.c_int, [_P, C.c_int, C.c_size_t, C.POINTER(_P)]),
    "jfgpu_parser_set_min_quality": (C.c_int, [_P, C.c_int]),
    "jfgpu_parser_last_ms": (C.c_int, [_P, C.POINTER(C.c_double)]),
    "jfgpu_set_growth": (C.c_int, [_P, C.c_int]),
    "jfgpu_reference_matrix": (C.c_int, [C.c_uint32, C.c_uint32, _P]),
    "jfgpu_table_bytes": (C.c_int, [C.c_uint32, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "jfgpu_set_spill": (C.c_int, [_P, _P, _P]),
    "jfgpu_set_operation": (C.c_int, [_P, C.c_int]),
    "jfgpu_set_mode": (C.c_int, [_P, C.c_int]),
    "jfgpu_reserve": (C.c_int, [_P, C.c_uint64]),
    "jfgpu_profile_enable": (C.c_int, [_P, C.c_int]),
    "jfgpu_profile_get": (C.c_int, [_P, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "jfgpu_profile_reset": (C.c_int, [_P]),
    "jfgpu_get_counters": (C.c_int, [_P, C.POINTER(C.c_uint64), C.c_uint32]),
    "jfgpu_gen_reads_dev": (C.c_int, [_P, _P, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64]),
    "jfgpu_gen_genome_reads_dev": (C.c_int, [_P, _P, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64, C.c_double, C.c_uint64]),
    "jfgpu_gups": (C.c_int, [_P, C.c_uint64, C.c_int, C.POINTER(C.c_double)]),
    "jfgpu_malloc_host": (C.c_int, [C.c_size_t, C.POINTER(_P)]),
    "jfgpu_free_host": (C.c_int, [_P]),
    "jfgpu_malloc_dev": (C.c_int, [_P, C.c_size_t, C.POINTER(_P)]),
    "jfgpu_free_dev": (C.c_int, [_P, _P]),
    "jfgpu_memcpy_h2d": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "jfgpu_memcpy_d2h": (C.c_int, [_P, _P, _P, C.c_size_t]),
}

_lib = None


def load():
    """Load libjfgpu.so.  Raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `make engine` "
                              "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def _check(rc):
    if rc != OK:
        raise JfgpuError(rc, load().jfgpu_last_error().decode(errors="replace"))


def reference_matrix(lsize, key_len):
    """The reference's default hash matrix for 2^lsize positions and key_len-bit keys (host only)."""
    cols = np.zeros(key_len, dtype=np.uint64)
    _check(load().jfgpu_reference_matrix(lsize, key_len, cols.ctypes.data))
    return cols


def device_count():
    return load().jfgpu_device_count()


def _ptr(x):
    """numpy array -> host pointer; int -> raw (device) pointer; torch tensor -> data_ptr()."""
    if x is None:
        return None
    if isinstance(x, np.ndarray):
        return x.ctypes.data
    if isinstance(x, int):
        return x
    if hasattr(x, "data_ptr"):
        return x.data_ptr()
    raise TypeError(type(x))


def xs_hash(key, r, key_bits=42):
    """kmer_core.hpp: xs_hash -- the xor-shift matrix family as a function (python ints, one-word keys)."""
    m32 = 0xFFFFFFFF
    y0 = key ^ (key >> 9) ^ (key >> 21) ^ ((key >> 32) if key_bits > 53 else 0)
    if r > 32:
        lo = y0 & m32
        lo ^= (lo << 13) & m32
        lo ^= (lo << 7) & m32
        lo ^= lo >> 17
        hi = ((key >> 32) ^ (lo >> 9)) & ((1 << (r - 32)) - 1)
        return (hi << 32) | lo
    m = (1 << r) - 1
    y = y0 & m
    y ^= (y << 13) & m
    y ^= (y << 7) & m
    y ^= y >> 17
    return y


def xs_hash_wide(lo, hi, r):
    """kmer_core.hpp: xs_hash_wide -- two-word keys folded to one word, then xs_hash."""
    m64 = (1 << 64) - 1
    return xs_hash(lo ^ hi ^ (((hi << 25) | (hi >> 39)) & m64), r, 64)


MATRIX_KINDS = {"default": 0, "xs": 1, "xorshift": 1, "reference": 2}      # include/jfgpu.h: JFGPU_MATRIX_*


class Table:
    """One hash shard in one GPU's HBM (jfgpu_table*)."""

    def __init__(self, k, size, canonical=True, device=-1, shard_bits=0, shard_id=0, matrix_seed=0,
                 matrix_columns=None, out_counter_len=4, matrix_kind=0):
        self._lib = load()
        p = Params()
        p.k, p.canonical, p.size, p.device = k, int(bool(canonical)), int(size), device
        p.shard_bits, p.shard_id, p.matrix_seed = shard_bits, shard_id, matrix_seed
        self._cols = None
        if matrix_columns is not None:
            self._cols = np.ascontiguousarray(matrix_columns, dtype=np.uint64)
            p.matrix_columns = self._cols.ctypes.data_as(C.POINTER(C.c_uint64))
        p.out_counter_len = out_counter_len
        p.matrix_kind = MATRIX_KINDS[matrix_kind] if isinstance(matrix_kind, str) else int(matrix_kind)
        h = _P()
        _check(self._lib.jfgpu_create(C.byref(p), C.byref(h)))
        self._h = h
        self.info = Info()
        _check(self._lib.jfgpu_get_info(self._h, C.byref(self.info)))
        self.k = k
        self.key_words = (2 * k + 63) // 64

    def matrix_is_xorshift(self):
        """Is the table's matrix the xor-shift family's member for its shape?  (restated here from jellyfish_amd/csrc/kmer_core.hpp:
        xs_hash, for reporting; the engine decides for itself)"""
        cols = self.matrix()
        r, c = int(self.info.lsize), int(self.info.key_len)
        if c > 128 or r >= c or r >= 64:
            return False
        img = lambda j: xs_hash(1 << j, r, c) if c <= 64 else (xs_hash_wide(1 << j, 0, r) if j < 64 else xs_hash_wide(0, 1 << (j - 64), r))
        return all(int(cols[c - 1 - j]) == img(j) for j in range(c))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.jfgpu_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- geometry
    def matrix(self):
        cols = np.zeros(2 * self.k, dtype=np.uint64)
        _check(self._lib.jfgpu_get_matrix(self._h, cols.ctypes.data))
        return cols

    def clear(self):
        _check(self._lib.jfgpu_clear(self._h))

    def sync(self):
        _check(self._lib.jfgpu_sync(self._h))
        _check(self._lib.jfgpu_get_info(self._h, C.byref(self.info)))     # the table may have doubled itself

    def wait(self):
        _check(self._lib.jfgpu_wait(self._h))

    # -- hot path
    def count_ascii(self, bases: bytes):
        buf = np.frombuffer(bases, dtype=np.uint8)
        _check(self._lib.jfgpu_count_ascii(self._h, buf.ctypes.data if len(bases) else None, len(bases)))

    def count_ascii_dev(self, d_ptr, n):
        _check(self._lib.jfgpu_count_ascii_dev(self._h, _ptr(d_ptr), n))

    def add_keys(self, keys, val=1, want_new=False):
        keys = np.ascontiguousarray(keys, dtype=np.uint64).reshape(-1, self.key_words)   # (n, words) little-endian words
        is_new = np.zeros(len(keys), dtype=np.uint8) if want_new else None
        _check(self._lib.jfgpu_add_keys(self._h, keys.ctypes.data, len(keys), val, _ptr(is_new)))
        return is_new

    def add_key_vals(self, keys, vals):
        keys = np.ascontiguousarray(keys, dtype=np.uint64).reshape(-1, self.key_words)
        vals = np.ascontiguousarray(vals, dtype=np.uint64)
        assert len(keys) == len(vals)
        _check(self._lib.jfgpu_add_key_vals(self._h, keys.ctypes.data, vals.ctypes.data, len(keys)))

    def add_keys_dev(self, d_keys, n, val=1, d_is_new=None):
        _check(self._lib.jfgpu_add_keys_dev(self._h, _ptr(d_keys), n, val, _ptr(d_is_new)))

    def lookup(self, keys):
        keys = np.ascontiguousarray(keys, dtype=np.uint64).reshape(-1, self.key_words)
        vals = np.zeros(len(keys), dtype=np.uint64)
        found = np.zeros(len(keys), dtype=np.uint8)
        _check(self._lib.jfgpu_lookup(self._h, keys.ctypes.data, len(keys), vals.ctypes.data, found.ctypes.data))
        return vals, found.astype(bool)

    def partition_ascii_dev(self, d_bases, n, d_keys_out, capacity):
        counts = np.zeros(1 << self.info.shard_bits, dtype=np.uint64)
        _check(self._lib.jfgpu_partition_ascii_dev(self._h, _ptr(d_bases), n, _ptr(d_keys_out), capacity, counts.ctypes.data))
        return counts

    # -- results
    def stats(self, lower=0, upper=2 ** 64 - 1):
        s = Stats()
        _check(self._lib.jfgpu_stats_compute(self._h, lower, upper, C.byref(s)))
        return s

    def digest(self, lower=0, upper=2 ** 64 - 1):
        """(records, sum of counts, sum of h, xor of h): order-independent checksum of the table's content."""
        out = np.zeros(4, dtype=np.uint64)
        _check(self._lib.jfgpu_digest(self._h, lower, upper, out.ctypes.data))
        return tuple(int(x) for x in out)

    def histo(self, low=1, high=10000, inc=1):
        """Same bucket arithmetic as histo_main.cc:60-66; returns (first_col, inc, counts)."""
        base = 0 if inc >= low else low - inc
        ceil = high + inc
        nb = (ceil + inc - base) // inc
        h = np.zeros(nb, dtype=np.uint64)
        _check(self._lib.jfgpu_histo(self._h, base, ceil, inc, h.ctypes.data, nb))
        return base, inc, h

    def dump_records(self, lower=0, upper=2 ** 64 - 1, chunk_records=1 << 20):
        """All records of this shard in (pos, key) order as one uint8 array (n, record_bytes)."""
        n = C.c_uint64()
        rb = C.c_uint32()
        _check(self._lib.jfgpu_dump_begin(self._h, lower, upper, C.byref(n), C.byref(rb)))
        try:
            chunk_records = max(chunk_records, self.info.tile_slots)
            out = np.zeros((n.value, rb.value), dtype=np.uint8)
            buf = np.zeros((chunk_records, rb.value), dtype=np.uint8)
            got = 0
            while True:
                nr = C.c_uint64()
                _check(self._lib.jfgpu_dump_next(self._h, buf.ctypes.data, chunk_records, C.byref(nr)))
                if nr.value == 0:
                    break
                out[got:got + nr.value] = buf[:nr.value]
                got += nr.value
            assert got == n.value, (got, n.value)
            return out
        finally:
            _check(self._lib.jfgpu_dump_end(self._h))

    def attach_bloom(self, bloom):
        """count --bc: admit only k-mers the Bloom counter has seen at least twice (None detaches)."""
        self._bloom = bloom
        _check(self._lib.jfgpu_attach_bloom(self._h, bloom._h if bloom is not None else None))

    def set_growth(self, on):
        _check(self._lib.jfgpu_set_growth(self._h, int(bool(on))))

    def set_spill(self, fn):
        """jfgpu_set_spill: fn() is called (table full, growth off) to write the table out; return 0 on success."""
        self._spill_cb = C.CFUNCTYPE(C.c_int, C.c_void_p)(lambda _u: int(fn() or 0)) if fn is not None else None
        _check(self._lib.jfgpu_set_spill(self._h, C.cast(self._spill_cb, C.c_void_p) if self._spill_cb else None, None))

    def refresh_info(self):
        _check(self._lib.jfgpu_get_info(self._h, C.byref(self.info)))
        return self.info

    def set_operation(self, op):
        """0 count, 1 prime (set), 2 update (count only what is already there): the passes of `count --if`."""
        _check(self._lib.jfgpu_set_operation(self._h, op))

    def set_mode(self, mode):
        """0 auto, 1 direct (global atomics), 2 partitioned (LDS tiles)."""
        _check(self._lib.jfgpu_set_mode(self._h, mode))

    def reserve(self, input_bytes):
        _check(self._lib.jfgpu_reserve(self._h, int(input_bytes)))

    # -- measurement helpers
    def profile_enable(self, on=True):
        _check(self._lib.jfgpu_profile_enable(self._h, int(on)))

    def profile_get(self, which):
        ms, ln, un = C.c_double(), C.c_uint64(), C.c_uint64()
        _check(self._lib.jfgpu_profile_get(self._h, which, C.byref(ms), C.byref(ln), C.byref(un)))
        return ms.value, ln.value, un.value

    def profile_reset(self):
        _check(self._lib.jfgpu_profile_reset(self._h))

    def profile_spans(self, cap=65536):
        """[(slot, device ms), ...] of every profiled launch since the last reset, in launch order (jfgpu_profile_spans)."""
        which = np.zeros(cap, dtype=np.int32)
        ms = np.zeros(cap, dtype=np.float64)
        n = C.c_size_t()
        _check(self._lib.jfgpu_profile_spans(self._h, which.ctypes.data, ms.ctypes.data, cap, C.byref(n)))
        k = min(n.value, cap)
        return list(zip(which[:k].tolist(), ms[:k].tolist()))

    COUNTER_NAMES = ("full", "mers", "ovf_full", "ovf_used", "misrouted", "direct", "t_items", "t_queued", "flushes_plain", "flushes_heavy",
                     "p2_roles", "p2_ring", "p2_sort", "p2_exact", "p1_ring", "p1_other")

    def counters(self):
        """jfgpu_get_counters: which paths the work since the last clear took."""
        out = (C.c_uint64 * len(self.COUNTER_NAMES))()
        _check(self._lib.jfgpu_get_counters(self._h, out, len(self.COUNTER_NAMES)))
        return dict(zip(self.COUNTER_NAMES, (int(x) for x in out)))

    def gen_reads_dev(self, d_out, first_read, n_reads, read_len, seed):
        _check(self._lib.jfgpu_gen_reads_dev(self._h, _ptr(d_out), first_read, n_reads, read_len, seed))

    def gen_genome_reads_dev(self, d_out, first_read, n_reads, read_len, genome_len, substitution_rate, seed):
        _check(self._lib.jfgpu_gen_genome_reads_dev(self._h, _ptr(d_out), first_read, n_reads, read_len, genome_len, substitution_rate, seed))

    def gups(self, n_updates, mode=0):
        v = C.c_double()
        _check(self._lib.jfgpu_gups(self._h, n_updates, mode, C.byref(v)))
        return v.value

    def malloc(self, nbytes):
        p = _P()
        _check(self._lib.jfgpu_malloc_dev(self._h, nbytes, C.byref(p)))
        return p.value

    def free(self, p):
        _check(self._lib.jfgpu_free_dev(self._h, p))

    def h2d(self, d_dst, src: np.ndarray):
        src = np.ascontiguousarray(src)
        _check(self._lib.jfgpu_memcpy_h2d(self._h, d_dst, src.ctypes.data, src.nbytes))

    def d2h(self, d_src, nbytes):
        out = np.zeros(nbytes, dtype=np.uint8)
        _check(self._lib.jfgpu_memcpy_d2h(self._h, out.ctypes.data, d_src, nbytes))
        return out


def comm_unique_id():
    """128 opaque bytes (ncclUniqueId) made on one rank and handed to the others before Comm(...)."""
    buf = np.zeros(128, dtype=np.uint8)
    _check(load().jfgpu_comm_unique_id(buf.ctypes.data))
    return bytes(buf)


class Comm:
    """The multi-GPU exchange (jfgpu_comm*): RCCL between one process per GPU, or `local=True`: all shards in this process."""

    def __init__(self, world, rank=0, unique_id=None, device=-1, local=False):
        self._lib = load()
        self.world, self.rank, self.local = world, rank, local
        h = _P()
        if local:
            _check(self._lib.jfgpu_comm_create_local(world, device, C.byref(h)))
        else:
            idb = np.frombuffer(unique_id, dtype=np.uint8).copy()
            assert len(idb) == 128
            _check(self._lib.jfgpu_comm_create(world, rank, idb.ctypes.data, device, C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.jfgpu_comm_destroy(self._h)
            self._h = None

    __del__ = close

    def step(self, table, d_ptr, n):
        """One collective step of this rank: route, exchange, insert what arrived for the previous step."""
        _check(self._lib.jfgpu_comm_count_ascii_dev(self._h, table._h, _ptr(d_ptr), n))

    def local_step(self, tables, d_ptrs, ns):
        w = self.world
        th = (_P * w)(*[t._h for t in tables])
        dp = (_P * w)(*[_ptr(p) for p in d_ptrs])
        nn = (C.c_size_t * w)(*ns)
        _check(self._lib.jfgpu_comm_local_step(self._h, th, dp, nn))

    def finish(self):
        s, r = C.c_uint64(), C.c_uint64()
        _check(self._lib.jfgpu_comm_finish(self._h, C.byref(s), C.byref(r)))
        return s.value, r.value

    def allreduce(self, values, op="sum"):
        """values (<= 64 integers) replaced by their sum / max over the ranks (RCCL transport; collective, synchronous)."""
        a = np.array(values, dtype=np.uint64)
        _check(self._lib.jfgpu_comm_allreduce_u64(self._h, a.ctypes.data, len(a), {"sum": 0, "max": 1}[op]))
        return a.tolist()

    def bc_merge(self, blooms):
        """jfgpu_comm_bc_merge(_local): the ranks' Bloom counters (one here, or all W of a local communicator, in rank order)
        become the counter of the whole input on every rank."""
        if self.local:
            arr = (_P * self.world)(*[b._h for b in blooms])
            _check(self._lib.jfgpu_comm_bc_merge_local(self._h, arr))
        else:
            _check(self._lib.jfgpu_comm_bc_merge(self._h, blooms._h))

    def world_rank(self):
        """(world, rank) as the communicator itself knows them (jfgpu_comm_world)."""
        w, r = C.c_int(), C.c_int()
        _check(self._lib.jfgpu_comm_world(self._h, C.byref(w), C.byref(r)))
        return w.value, r.value

    def exchange_times(self, cap=65536):
        """[(device ms, bytes this rank sent over the wires), ...] of every exchange since the last call (jfgpu_comm_exchange_times)."""
        ms = np.zeros(cap, dtype=np.float64)
        by = np.zeros(cap, dtype=np.uint64)
        n = C.c_size_t()
        _check(self._lib.jfgpu_comm_exchange_times(self._h, ms.ctypes.data, by.ctypes.data, cap, C.byref(n)))
        k = min(n.value, cap)
        return list(zip(ms[:k].tolist(), by[:k].tolist()))

    def allgather(self, mine):
        a = np.zeros(self.world, dtype=np.uint64)
        _check(self._lib.jfgpu_comm_allgather_u64(self._h, int(mine), a.ctypes.data))
        return a.tolist()


class Bloom:
    """Bloom counter of `jellyfish bc` in HBM (jfgpu_bloom*)."""

    def __init__(self, k, m, nb_hashes, canonical=True, device=-1, seed=0, matrix1=None, matrix2=None, one_pass_filter=False):
        self._lib = load()
        p = BloomParams()
        p.k, p.canonical, p.m, p.nb_hashes, p.device, p.seed = k, int(bool(canonical)), int(m), int(nb_hashes), device, seed
        self._m1 = self._m2 = None
        if matrix1 is not None:
            self._m1 = np.ascontiguousarray(matrix1, dtype=np.uint64)
            self._m2 = np.ascontiguousarray(matrix2, dtype=np.uint64)
            p.matrix1 = self._m1.ctypes.data_as(C.POINTER(C.c_uint64))
            p.matrix2 = self._m2.ctypes.data_as(C.POINTER(C.c_uint64))
        h = _P()
        _check((self._lib.jfgpu_bf_create if one_pass_filter else self._lib.jfgpu_bc_create)(C.byref(p), C.byref(h)))
        self._h = h
        self.k = k
        m_, nh, nbytes = C.c_uint64(), C.c_uint32(), C.c_uint64()
        self.matrix1 = np.zeros(2 * k, dtype=np.uint64)
        self.matrix2 = np.zeros(2 * k, dtype=np.uint64)
        _check(self._lib.jfgpu_bc_get_info(self._h, C.byref(m_), C.byref(nh), C.byref(nbytes), self.matrix1.ctypes.data, self.matrix2.ctypes.data))
        self.m, self.nb_hashes, self.nb_bytes = m_.value, nh.value, nbytes.value

    def close(self):
        if getattr(self, "_h", None):
            self._lib.jfgpu_bc_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def insert_ascii(self, bases: bytes):
        buf = np.frombuffer(bases, dtype=np.uint8)
        _check(self._lib.jfgpu_bc_insert_ascii(self._h, buf.ctypes.data if len(bases) else None, len(bases)))

    def insert_ascii_dev(self, d_ptr, n):
        _check(self._lib.jfgpu_bc_insert_ascii_dev(self._h, _ptr(d_ptr), n))

    def sync(self):
        n = C.c_uint64()
        _check(self._lib.jfgpu_bc_sync(self._h, C.byref(n)))
        return n.value

    def clear(self):
        _check(self._lib.jfgpu_bc_clear(self._h))

    def read(self):
        out = np.zeros(self.nb_bytes, dtype=np.uint8)
        _check(self._lib.jfgpu_bc_read(self._h, out.ctypes.data))
        return out

    def load(self, data):
        data = np.ascontiguousarray(data, dtype=np.uint8)
        assert len(data) == self.nb_bytes
        _check(self._lib.jfgpu_bc_load(self._h, data.ctypes.data))

    def set_mode(self, mode):
        """0 auto, 1 direct (global compare-and-swap per cell), 2 partitioned (cell updates applied per 64 KiB segment in LDS)."""
        _check(self._lib.jfgpu_bc_set_mode(self._h, mode))

    def reserve(self, workspace_bytes):
        _check(self._lib.jfgpu_bc_reserve(self._h, int(workspace_bytes)))

    def profile_enable(self, on=True):
        _check(self._lib.jfgpu_bc_profile_enable(self._h, int(on)))

    def profile_get(self, which):
        ms, ln, un = C.c_double(), C.c_uint64(), C.c_uint64()
        _check(self._lib.jfgpu_bc_profile_get(self._h, which, C.byref(ms), C.byref(ln), C.byref(un)))
        return ms.value, ln.value, un.value

    def profile_reset(self):
        _check(self._lib.jfgpu_bc_profile_reset(self._h))

    def ring_p2_launches(self):
        """Launches of the ring P2 kernel (p2_ring_kernel<BloomRingDirect>) since the last profile_reset: profile slot 4."""
        return self.profile_get(4)[1]

    def keys(self, keys, insert=False):
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        out = np.zeros(len(keys), dtype=np.uint8)
        _check(self._lib.jfgpu_bc_keys(self._h, keys.ctypes.data, len(keys), out.ctypes.data, int(insert)))
        return out


def opt_m(fp, n):
    return load().jfgpu_bc_opt_m(fp, n)


def opt_k(fp):
    return load().jfgpu_bc_opt_k(fp)


def digest_of(keys, counts):
    """The same checksum from decoded records (keys: (n,) or (n, words) uint64, counts: (n,) uint64), in numpy."""
    def mix(z):
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    if keys.ndim == 1:
        keys = keys.reshape(-1, 1)
    counts = np.ascontiguousarray(counts, dtype=np.uint64)
    with np.errstate(over="ignore"):
        h = np.full(len(keys), 0x9E3779B97F4A7C15, dtype=np.uint64)
        for w in range(keys.shape[1]):
            h = mix(h ^ keys[:, w])
        h = mix(h ^ counts)
        return (len(keys), int(counts.sum(dtype=np.uint64)), int(h.sum(dtype=np.uint64)),
                int(np.bitwise_xor.reduce(h)) if len(h) else 0)


def decode_records(recs: np.ndarray, k: int, counter_len: int):
    """(n, record_bytes) uint8 -> (keys, counts uint64 (n,)); keys is uint64 (n,) for k <= 32 and
    (n, ceil(k/32)) little-endian words otherwise (binary_dumper.hpp:36-40 layout: ceil(2k/8) key
    bytes LE, then counter_len bytes LE)."""
    kb = (2 * k + 7) // 8
    kw = (2 * k + 63) // 64
    n = len(recs)
    keys = np.zeros((n, kw), dtype=np.uint64)
    cnts = np.zeros(n, dtype=np.uint64)
    for b in range(kb):
        keys[:, b // 8] |= recs[:, b].astype(np.uint64) << np.uint64(8 * (b % 8))
    for b in range(counter_len):
        cnts |= recs[:, kb + b].astype(np.uint64) << np.uint64(8 * b)
    return (keys[:, 0] if kw == 1 else keys), cnts


class Parser:
    """FASTA / FASTQ bytes -> contract buffer on the device (jfgpu_parser*)."""

    def __init__(self, k, device=-1):
        self._lib = load()
        h = _P()
        _check(self._lib.jfgpu_parser_create(device, k, C.byref(h)))
        self._h = h
        self.records = 0

    def close(self):
        if getattr(self, "_h", None):
            self._lib.jfgpu_parser_destroy(self._h)
            self._h = None

    __del__ = close

    def _call(self, fn, ptr, n, flags):
        out, n_out, recs = _P(), C.c_size_t(), C.c_uint64()
        _check(fn(self._h, ptr, n, flags, C.byref(out), C.byref(n_out), C.byref(recs)))
        self.records = recs.value
        return (out.value or 0), n_out.value

    def parse(self, data: bytes, flags):
        """Host bytes in; (device pointer, length) of the contract buffer out."""
        return self._call(self._lib.jfgpu_parser_parse, data, len(data), flags)

    def parse_dev(self, d_ptr, n, flags):
        return self._call(self._lib.jfgpu_parser_parse_dev, _ptr(d_ptr), n, flags)

    def set_min_quality(self, ch):
        _check(self._lib.jfgpu_parser_set_min_quality(self._h, int(ch)))

    def last_ms(self):
        ms = C.c_double()
        _check(self._lib.jfgpu_parser_last_ms(self._h, C.byref(ms)))
        return ms.value
