"""jellyfish_amd -- MI355X-native engine for the `jellyfish count` hot path.

Layout (only what the path needs):
  csrc/      hand-written HIP kernels for gfx950 + the C ABI (include/jfgpu.h)
  lib/       built libjfgpu.so (git-ignored; `make engine`)
  include/   C++ facade mirroring the reference API (mer_dna, hash_counter, file_header, dumpers)
  cli/       `jellyfish-amd count|dump|histo|stats|query|info` host program
  capi.py    ctypes plumbing over the C ABI for tests/ and bench.py (multi-GPU: jfgpu_comm_* under the same ABI)
  compat/    namespace jellyfish headers the reference's client sources compile against
"""
from . import capi  # noqa: F401

__all__ = ["capi"]
