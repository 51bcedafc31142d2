"""ctypes binding of the parity oracle (oracle/jf_oracle.c) -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "_build", "libjf_oracle.so")
REF_JF = os.path.join(ORACLE_DIR, "_ref", "ref_jf")
REF_GEN = os.path.join(ORACLE_DIR, "_ref", "ref_generate_sequence")


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "restatement"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(os.path.join(ORACLE_DIR, "jf_oracle.c")):
            build()
        L = C.CDLL(LIB)
        L.jfo_code.argtypes = [C.c_ubyte]; L.jfo_code.restype = C.c_int
        L.jfo_nb_words.argtypes = [C.c_uint]; L.jfo_nb_words.restype = C.c_uint
        L.jfo_parse_file.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t]; L.jfo_parse_file.restype = C.c_size_t
        L.jfo_extract.argtypes = [C.c_void_p, C.c_size_t, C.c_uint, C.c_int, C.c_void_p, C.c_size_t]; L.jfo_extract.restype = C.c_size_t
        L.jfo_sort_count.argtypes = [C.c_void_p, C.c_size_t, C.c_uint, C.c_void_p, C.c_void_p]; L.jfo_sort_count.restype = C.c_size_t
        L.jfo_matrix_times.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_void_p]; L.jfo_matrix_times.restype = C.c_uint64
        L.jfo_revcomp.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
        L.jfo_to_str.argtypes = [C.c_void_p, C.c_uint, C.c_char_p]
        L.jfo_from_str.argtypes = [C.c_char_p, C.c_uint, C.c_void_p]; L.jfo_from_str.restype = C.c_int
        L.jfo_bc_insert.argtypes = [C.c_void_p, C.c_uint64, C.c_uint, C.c_uint64, C.c_uint64]; L.jfo_bc_insert.restype = C.c_uint
        L.jfo_bc_check.argtypes = [C.c_void_p, C.c_uint64, C.c_uint, C.c_uint64, C.c_uint64]; L.jfo_bc_check.restype = C.c_uint
        _lib = L
    return _lib


def nb_words(k):
    return (2 * k + 63) // 64


def parse_file(data: bytes) -> bytes:
    """FASTA/FASTQ file bytes -> parser-contract buffer ('N' between records)."""
    out = C.create_string_buffer(len(data) + 1)
    n = lib().jfo_parse_file(data, len(data), out, len(data) + 1)
    if n == C.c_size_t(-1).value:
        raise RuntimeError("Unsupported format / invalid fastq")
    return out.raw[:n]


def extract(seq: bytes, k: int, canonical: bool) -> np.ndarray:
    """All (canonical) k-mers of a contract buffer, in order. shape (n, nb_words)."""
    nw = nb_words(k)
    cap = max(1, len(seq))
    out = np.zeros((cap, nw), dtype=np.uint64)
    buf = np.frombuffer(seq, dtype=np.uint8)
    n = lib().jfo_extract(buf.ctypes.data if len(seq) else None, len(seq), k, int(canonical), out.ctypes.data, cap)
    return out[:n]


def count(seq: bytes, k: int, canonical: bool):
    """Exact {k-mer -> count}: returns (keys (d, nw) uint64 sorted numerically, counts (d,) uint64)."""
    kmers = np.ascontiguousarray(extract(seq, k, canonical))
    n, nw = kmers.shape
    keys = np.zeros((max(n, 1), nw), dtype=np.uint64)
    counts = np.zeros(max(n, 1), dtype=np.uint64)
    d = lib().jfo_sort_count(kmers.ctypes.data, n, nw, keys.ctypes.data, counts.ctypes.data)
    return keys[:d], counts[:d]


def matrix_times(columns, r, c, keys: np.ndarray) -> np.ndarray:
    """pos = M * key for each row of keys (n, nw). columns=None => identity."""
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    if keys.ndim == 1:
        keys = keys.reshape(-1, 1)
    cols = None if columns is None else np.ascontiguousarray(columns, dtype=np.uint64)
    out = np.zeros(len(keys), dtype=np.uint64)
    L = lib()
    for i in range(len(keys)):
        out[i] = L.jfo_matrix_times(None if cols is None else cols.ctypes.data, r, c, keys[i].ctypes.data)
    return out


def to_str(key, k):
    key = np.ascontiguousarray(key, dtype=np.uint64).reshape(-1)
    out = C.create_string_buffer(k + 1)
    lib().jfo_to_str(key.ctypes.data, k, out)
    return out.value.decode()


def from_str(s, k):
    key = np.zeros(nb_words(k), dtype=np.uint64)
    ok = lib().jfo_from_str(s.encode(), k, key.ctypes.data)
    if not ok:
        raise ValueError("invalid mer " + s)
    return key


def have_ref():
    return os.access(REF_JF, os.X_OK)


def ref_count_dump(fasta_path, k, size, canonical=True, threads=1, workdir=None, extra=()):
    """Run the REFERENCE (oracle/_ref/ref_jf) on a file; return sorted `dump -c` lines."""
    import tempfile
    d = workdir or tempfile.mkdtemp(prefix="jfref")
    out = os.path.join(d, "ref.jf")
    cmd = [REF_JF, "count", "-m", str(k), "-s", str(size), "-t", str(threads), "-o", out]
    if canonical:
        cmd.append("-C")
    cmd += list(extra) + [fasta_path]
    subprocess.check_call(cmd)
    txt = subprocess.check_output([REF_JF, "dump", "-c", out]).decode()
    return sorted(txt.splitlines()), out
