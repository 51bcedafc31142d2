/* include/jfgpu.h -- C ABI of the MI355X-native k-mer counting engine (libjfgpu.so).
 *
 * This is the drop-in boundary for the `jellyfish count` hot path.  The
 * reference (gmarcais/Jellyfish, /root/reference) has no FFI: its seam is the C++
 * template API `jellyfish::cooperative::hash_counter<mer_dna>` plus the
 * binary/sorted file format.  Each entry point below names the reference
 * interface (file:line relative to /root/reference) it stands in for; the C++
 * facade in jellyfish_amd/include/jellyfish_amd/ and the `jellyfish-amd` CLI sit
 * on top of exactly these functions.
 *
 * Conventions
 *  - plain C: opaque handle, pointers + sizes, int return codes (0 = ok), no
 *    exceptions cross the boundary; jfgpu_last_error() gives the message the
 *    reference would have thrown (e.g. "Hash full", hash_counter.hpp:194-195).
 *  - k-mers are arrays of ceil(k/32) little-endian uint64 words, word 0 least
 *    significant, base i of the string (0 = leftmost) at bits 2(k-1-i), A=0 C=1
 *    G=2 T=3 -- the in-memory layout of mer_dna (mer_dna.hpp:143-155,526-542),
 *    i.e. what mer_dna::data() returns and binary_writer writes.
 *  - buffers are host pointers unless the function name ends in _dev (then the
 *    pointer is device memory on the table's GPU, e.g. a torch tensor's
 *    data_ptr()).  All work is enqueued on the table's HIP stream; functions that
 *    return results to the host synchronise that stream.
 *  - the engine needs a gfx950 GPU.  There is no CPU fallback: every call fails
 *    with JFGPU_E_NO_DEVICE when no HIP device is usable.
 */
#ifndef JFGPU_H
#define JFGPU_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JFGPU_ABI_VERSION 1

enum {
  JFGPU_OK = 0,
  JFGPU_E_INVALID = 1,    /* bad argument (std::length_error / out_of_range in the reference) */
  JFGPU_E_NO_DEVICE = 2,  /* no usable HIP device / kernel image */
  JFGPU_E_ALLOC = 3,      /* large_hash::array::ErrorAllocation, large_hash_array.hpp:55,169-172 */
  JFGPU_E_FULL = 4,       /* std::runtime_error("Hash full"), hash_counter.hpp:194-195 */
  JFGPU_E_HIP = 5,        /* HIP runtime error */
  JFGPU_E_UNSUPPORTED = 6,/* combination not built (e.g. k > 128, a sharded table for k > 64, --bc / --if over shards of two-word keys) */
  JFGPU_E_FORMAT = 7      /* device parser: chunk is not in the strict layout it handles; give it to the host parser */
};

typedef struct jfgpu_table jfgpu_table; /* opaque: one hash shard resident in one GPU's HBM */

/* Parameters of a table.  Mirrors the ctor of hash_counter / large_hash::array
 * (hash_counter.hpp:56-64, large_hash_array.hpp:992-1001) minus the CPU-only knobs
 * (nb_threads, reprobe schedule: the in-memory probing is never serialised). */
typedef struct jfgpu_params {
  uint32_t k;            /* mer length; key_len = 2k bits.  1..128 (one to four 64-bit key words; the reference's
                            mer_dna takes any length, mer_dna.hpp:711-717 -- longer is JFGPU_E_UNSUPPORTED here) */
  uint32_t canonical;    /* count_main.cc -C: count min(mer, revcomp) */
  uint64_t size;         /* requested GLOBAL number of slots (hint: rounded up to a power of
                            two, capped at 4^k, raised to the engine minimum for large k) */
  int32_t  device;       /* HIP device ordinal, -1 = current device */
  uint32_t shard_bits;   /* log2(number of shards).  Shard s owns global positions whose top
                            shard_bits bits equal s; 0 = single GPU */
  uint32_t shard_id;     /* which shard this table is */
  uint64_t matrix_seed;  /* seed of the random GF(2) hash matrix; 0 = engine default.
                            All shards of one job must use the same seed */
  const uint64_t* matrix_columns; /* optional explicit matrix: 2k columns in file-header order
                            (file_header.hpp:35-64), r = log2(global size) rows; its low r x r
                            block must be invertible.  NULL = random from matrix_seed */
  uint32_t out_counter_len; /* bytes per count in dumps (binary_dumper ctor val_len,
                            count_main_cmdline.yaggo --out-counter-len); 0 = 4 */
  uint32_t matrix_kind;  /* which family the matrix comes from when neither matrix_columns nor matrix_seed is given:
                            JFGPU_MATRIX_DEFAULT (0, what the JFGPU_MATRIX environment switch says, else the reference's),
                            JFGPU_MATRIX_XORSHIFT or JFGPU_MATRIX_REFERENCE.  (Was `reserved`, always 0.) */
} jfgpu_params;

/* Matrix families.  The file format admits any r x 2k matrix with an invertible low r x r block -- readers take the
 * columns from the header (include/jellyfish/file_header.hpp:35-64) and multiply
 * (rectangular_binary_matrix.hpp:155-164) -- so the choice changes where a k-mer lands, never what is counted.
 *   REFERENCE: the matrix `jellyfish count` itself draws (lib/rectangular_binary_matrix.cc:160-247 on glibc's unseeded
 *              random()): file bodies byte-identical to the reference's, files mergeable with the reference's.
 *   XORSHIFT:  a fixed matrix per (size, k) made of xor-shift steps (jellyfish_amd/csrc/kmer_core.hpp: xs_hash), which
 *              the partition kernel evaluates in registers instead of six LDS table reads per k-mer.  One-word keys
 *              (k <= 32); longer keys get the reference family. */
enum { JFGPU_MATRIX_DEFAULT = 0, JFGPU_MATRIX_XORSHIFT = 1, JFGPU_MATRIX_REFERENCE = 2 };

/* Geometry + matrix actually used, for file_header::update_from_ary (file_header.hpp:25-33). */
typedef struct jfgpu_info {
  uint32_t k, key_len;       /* key_len = 2k */
  uint32_t canonical;
  uint32_t lsize;            /* log2(global size) = rows of the matrix */
  uint64_t size;             /* global number of slots (power of two) -> header "size" */
  uint64_t local_size;       /* slots held by this shard */
  uint32_t shard_bits, shard_id;
  uint32_t val_len;          /* bits of the in-slot count field -> header "val_len" */
  uint32_t slot_bytes;       /* 4 or 8 (one-word keys), 16 (two words), 32 (three and four) */
  uint32_t tile_slots;       /* probe domain (slots) */
  uint32_t matrix_identity;  /* 1 when size == 4^k (large_hash_array.hpp:997-1000) */
  uint32_t out_counter_len;
  uint32_t max_reprobe;      /* informational header fields (merge_files.cc:128,145-146) */
  uint64_t table_bytes;
} jfgpu_info;

typedef struct jfgpu_stats {
  uint64_t unique;     /* k-mers with count 1            (stats_main.cc:41) */
  uint64_t distinct;   /* number of different k-mers     (:44) */
  uint64_t total;      /* sum of counts                  (:42) */
  uint64_t max_count;  /*                                (:43) */
  uint64_t occupied;   /* slots in use (== distinct) */
  uint64_t mers_fed;   /* k-mer occurrences accepted by jfgpu_count_* so far */
} jfgpu_stats;

const char* jfgpu_last_error(void);       /* thread-local message of the last failing call */
int  jfgpu_abi_version(void);
int  jfgpu_device_count(void);            /* usable HIP devices (0 => nothing will work) */

/* hash_counter ctor / dtor (hash_counter.hpp:56-68). */
int  jfgpu_create(const jfgpu_params* params, jfgpu_table** out);
void jfgpu_destroy(jfgpu_table* t);
int  jfgpu_get_info(const jfgpu_table* t, jfgpu_info* out);
/* matrix() (large_hash_array.hpp:211): 2k columns, file-header order. */
int  jfgpu_get_matrix(const jfgpu_table* t, uint64_t* columns /* [2k] */);
/* array::clear() (large_hash_array.hpp:224-226). */
int  jfgpu_clear(jfgpu_table* t);
/* Block until everything enqueued so far has retired; reports a deferred
 * "Hash full" (the device cannot throw mid-kernel). == every thread called
 * hash_counter::done() (hash_counter.hpp:169-172). */
int  jfgpu_sync(jfgpu_table* t);
/* Wait until the kernels enqueued so far have finished WITHOUT applying pending partitioned
 * batches: after it returns, device buffers handed to jfgpu_count_ascii_dev / jfgpu_add_keys_dev
 * may be reused (their k-mers have been copied into the engine's workspace). */
int  jfgpu_wait(jfgpu_table* t);

/* ---- the hot path ------------------------------------------------------ */
/* mer_counter_base::start COUNT loop (sub_commands/count_main.cc:152-163) over one
 * parser-contract buffer (mer_overlap_sequence_parser.hpp:161-185: sequence
 * characters, any byte outside [ACGTacgt] resets the window, k-mers do not span
 * calls): encode (mer_iterator.hpp:53-81) -> canonical (:51) -> hash
 * (rectangular_binary_matrix.hpp:155-164) -> insert/increment
 * (large_hash_array.hpp:291-295,509-597,741-752).  Asynchronous. */
int  jfgpu_count_ascii_dev(jfgpu_table* t, const char* d_bases, size_t n);
int  jfgpu_count_ascii(jfgpu_table* t, const char* bases, size_t n);      /* host buffer (staged H2D) */

/* hash_counter::add(key, val) for a batch of already-encoded k-mers
 * (hash_counter.hpp:122-126; SWIG HashCounter.add swig/hash_counter.i:13-27).
 * is_new (optional, n bytes) receives 1 where the key was not present. */
int  jfgpu_add_keys_dev(jfgpu_table* t, const uint64_t* d_keys, size_t n, uint64_t val, uint8_t* d_is_new);
int  jfgpu_add_keys(jfgpu_table* t, const uint64_t* keys, size_t n, uint64_t val, uint8_t* is_new);

/* The same with a value per key: hash_counter::add(const mer_dna&, uint64_t val) over a batch (hash_counter.hpp:122-126 ->
 * large_hash_array.hpp:741-752 add_val) -- what loads the records of a binary/sorted file back into a table, so that
 * `query -s` (sub_commands/query_main.cc:44-51) answers from the device.  Host arrays; keys of one or two words. */
int  jfgpu_add_key_vals(jfgpu_table* t, const uint64_t* keys, const uint64_t* vals, size_t n);

/* array::get_val_for_key (large_hash_array.hpp:354-372) for a batch.  vals[i] = 0 and
 * found[i] = 0 when absent.  Keys must already be canonical if the table is. */
int  jfgpu_lookup_dev(jfgpu_table* t, const uint64_t* d_keys, size_t n, uint64_t* d_vals, uint8_t* d_found);
int  jfgpu_lookup(jfgpu_table* t, const uint64_t* keys, size_t n, uint64_t* vals, uint8_t* found);

/* ---- multi-GPU: hash-prefix partition --------------------- */
/* Encode + canonicalise + hash one contract buffer and bucket the k-mers by owning
 * shard instead of inserting them.  d_keys_out has room for `capacity` keys; shard
 * s's keys are written contiguously at d_keys_out + offsets[s] ... in arbitrary
 * order, with counts[s] of them; offsets are the exclusive prefix sum of counts
 * (so the buffer is densely packed and ready for an all-to-all-v).
 * counts_out: host array [1 << shard_bits].  Synchronous. */
int  jfgpu_partition_ascii_dev(jfgpu_table* t, const char* d_bases, size_t n,
                               uint64_t* d_keys_out, size_t capacity, uint64_t* counts_out);

/* The exchange itself: one process per GPU, shard r = the table positions whose top shard_bits bits are
 * r.  jfgpu_comm_count_ascii_dev is one step of a rank: route this rank's contract buffer by owner, exchange (RCCL
 * ncclSend / ncclRecv over xGMI, one group per round of at most 1 GiB per peer), insert what arrived -- pipelined by
 * one step, so the keys of step i travel while step i+1 is routed and step i-1 is inserted.  Collective: every rank
 * calls it the same number of times (n may be 0), then jfgpu_comm_finish, then reads its table (jfgpu_sync, dump...);
 * the shards' sorted dumps concatenated in rank order are the globally (pos, key)-sorted file body.
 *   id128: from jfgpu_comm_unique_id on one rank, handed to the others by whatever launched them.
 * jfgpu_comm_create_local: the same code with all `world` shards in ONE process on one device and device copies as
 * the transport (no RCCL): exists so that the sharded path can be tested on a single GPU. */
typedef struct jfgpu_comm jfgpu_comm;
int  jfgpu_comm_unique_id(uint8_t* id128);
int  jfgpu_comm_create(int world, int rank, const uint8_t* id128, int device, jfgpu_comm** out);
int  jfgpu_comm_create_local(int world, int device, jfgpu_comm** out);
void jfgpu_comm_destroy(jfgpu_comm* c);
int  jfgpu_comm_count_ascii_dev(jfgpu_comm* c, jfgpu_table* t, const char* d_bases, size_t n);
int  jfgpu_comm_local_step(jfgpu_comm* c, jfgpu_table** tables, const char* const* d_bases, const size_t* n);
int  jfgpu_comm_finish(jfgpu_comm* c, uint64_t* sent, uint64_t* received);
/* What a launcher has to agree on around the steps (RCCL transport, synchronous, collective): values[n] (n <= 64) replaced
 * by their sum (op 0) or maximum (op 1) over the ranks -- "does any rank still have input?", so that every rank makes the
 * same number of steps; all[world] = every rank's `mine` -- the records each shard will write, i.e. a rank's offset in
 * the common binary/sorted file (sorted_dumper.hpp:57-101 writes one file; here shard dumps are concatenated in rank order). */
int  jfgpu_comm_allreduce_u64(jfgpu_comm* c, uint64_t* values, int n, int op);
int  jfgpu_comm_allgather_u64(jfgpu_comm* c, uint64_t mine, uint64_t* all);
int  jfgpu_comm_world(const jfgpu_comm* c, int* world, int* rank);
/* Exchange timing (measurement helper): device time of every exchange since the communicator was created or this was last
 * called -- from the moment the routed data of a step was ready on the exchange stream to the moment everything of that
 * step had been sent and received -- and the bytes this rank put on the wires for it (its own share does not travel).
 * Waits for the exchange stream.  *n = exchanges recorded; at most cap entries are written; the log is cleared. */
int  jfgpu_comm_exchange_times(jfgpu_comm* c, double* ms, uint64_t* wire_bytes, size_t cap, size_t* n);

/* ---- results path ------------------------------------------------------ */
int  jfgpu_stats_compute(jfgpu_table* t, uint64_t lower, uint64_t upper, jfgpu_stats* out);
/* Order-independent checksum of the {k-mer -> count} content restricted to lower <= count <= upper:
 * out4 = { records, sum of counts, sum of h, xor of h } (mod 2^64) with h = mix(..mix(mix(S ^ w0) ^ w1).. ^ count),
 * mix = the splitmix64 finaliser, S = 0x9E3779B97F4A7C15, w = the key's little-endian words (mer_dna::data()).
 * No reference counterpart: it exists so that a 10 Gbp run is compared with the reference's table (the oracle driver
 * computes the same numbers from large_hash_array's iterators) without writing and sorting ~86 GB of records. */
int  jfgpu_digest(jfgpu_table* t, uint64_t lower, uint64_t upper, uint64_t* out4);
/* histo_main.cc:34-45: histo[0] counts vals < base, histo[n-1] vals > ceil,
 * else histo[(val-base)/inc]. */
int  jfgpu_histo(jfgpu_table* t, uint64_t base, uint64_t ceil, uint64_t inc, uint64_t* histo, uint64_t nb_buckets);
/* sorted_dumper + binary_writer (sorted_dumper.hpp:57-101, binary_dumper.hpp:36-40):
 * produce this shard's records in ascending (pos, key) order, fixed width
 * ceil(2k/8) key bytes + out_counter_len count bytes (saturated), filtered to
 * lower <= count <= upper.  _begin returns the total record count; _next streams
 * the records in order into a host buffer of `capacity_records` records (at least
 * one tile = jfgpu_info.tile_slots), *n_read == 0 marks the end; whole tiles are
 * sorted on the device, so every chunk is a contiguous piece of the file body. */
int  jfgpu_dump_begin(jfgpu_table* t, uint64_t lower, uint64_t upper, uint64_t* n_records, uint32_t* record_bytes);
int  jfgpu_dump_next(jfgpu_table* t, void* out, uint64_t capacity_records, uint64_t* n_read);
int  jfgpu_dump_end(jfgpu_table* t);

/* ---- Bloom counter: `jellyfish bc` and `count --bc` (BASELINE config 3) ------------------- */
typedef struct jfgpu_bloom jfgpu_bloom;   /* opaque: ceil(m/5) bytes of base-3 cells in HBM */
typedef struct jfgpu_bloom_params {
  uint32_t k, canonical;
  uint64_t m;                 /* number of cells (bloom_base::m(); header "size") */
  uint32_t nb_hashes;         /* header "nb_hashes" */
  int32_t  device;            /* -1 = current */
  uint64_t seed;              /* for the two random 64 x 2k matrices; 0 = default */
  const uint64_t* matrix1;    /* optional explicit matrices, 2k columns each, file-header order */
  const uint64_t* matrix2;    /*   (what load_bloom_filter reads back, count_main.cc:191-206) */
} jfgpu_bloom_params;
/* bloom_base::opt_m / opt_k (bloom_common.hpp:61-66) */
uint64_t jfgpu_bc_opt_m(double fp, uint64_t n);
uint32_t jfgpu_bc_opt_k(double fp);
/* mer_dna_bloom_counter ctor (bc_main.cc:114-120) / dtor */
int  jfgpu_bc_create(const jfgpu_bloom_params* p, jfgpu_bloom** out);
void jfgpu_bc_destroy(jfgpu_bloom* b);
/* mer_bloom_counter::start (bc_main.cc:67-71): filter.insert(*mers) for every k-mer of a contract buffer */
int  jfgpu_bc_insert_ascii_dev(jfgpu_bloom* b, const char* d_bases, size_t n);
int  jfgpu_bc_insert_ascii(jfgpu_bloom* b, const char* bases, size_t n);
int  jfgpu_bc_sync(jfgpu_bloom* b, uint64_t* mers_fed);
int  jfgpu_bc_clear(jfgpu_bloom* b);      /* all cells back to 0 (a fresh mer_dna_bloom_counter), k-mer tally reset */
int  jfgpu_bc_get_info(const jfgpu_bloom* b, uint64_t* m, uint32_t* nb_hashes, uint64_t* nb_bytes,
                       uint64_t* matrix1 /* [2k] or NULL */, uint64_t* matrix2);
/* write_bits / the istream ctor: the raw ceil(m/5) bytes of a "bloomcounter" file body */
int  jfgpu_bc_read(jfgpu_bloom* b, uint8_t* out);
int  jfgpu_bc_load(jfgpu_bloom* b, const uint8_t* data);
/* bloom_base::check / insert on encoded k-mers (query_main.cc Bloom branch); out[i] = 0, 1 or 2 */
int  jfgpu_bc_keys(jfgpu_bloom* b, const uint64_t* keys, size_t n, uint8_t* out, int do_insert);
/* count --bf-size N --bf-fp F (count_main.cc:121-131,321-323; bloom_filter.hpp:44-68): a Bloom filter of m = opt_m(F, N)
 * BITS with opt_k(F) hashes, filled by the count itself -- attached with jfgpu_attach_bloom, every k-mer sets its bits
 * and is counted only if all of them were set before (its first sighting only marks it).  Same params struct; the
 * result is order dependent in the reference too (tests/bloom_filter.sh bounds it statistically). */
int  jfgpu_bf_create(const jfgpu_bloom_params* p, jfgpu_bloom** out);
/* How jfgpu_bc_insert_* applies the increments (no reference counterpart; the array is the same either way because the
 * increments saturate and commute): 0 auto, 1 direct (one global compare-and-swap per cell), 2 partitioned (cell updates
 * routed to 64 KiB segments of the array and applied in LDS at the next jfgpu_bc_sync / _read / _keys / attach).
 * jfgpu_bc_reserve sizes the routing workspace up front (default: taken from free memory at the first large batch). */
int  jfgpu_bc_set_mode(jfgpu_bloom* b, int mode);
int  jfgpu_bc_reserve(jfgpu_bloom* b, uint64_t workspace_bytes);
/* per-stage device time on the counter's stream (bench.py): which = 0 direct, 1 route (P1), 2 partition (P2), 3 segments;
 * 4: launches (no time) of the ring kernel inside stage 2, so that a test can tell which P2 kernel a flush took */
int  jfgpu_bc_profile_enable(jfgpu_bloom* b, int on);
int  jfgpu_bc_profile_get(jfgpu_bloom* b, int which, double* ms, uint64_t* launches, uint64_t* units);
int  jfgpu_bc_profile_reset(jfgpu_bloom* b);
/* count --bc: from now on jfgpu_count_* admits a k-mer only if check(m) > 1 (count_main.cc:115-118).
 * b == NULL detaches.  The Bloom counter must outlive its use. */
int  jfgpu_attach_bloom(jfgpu_table* t, jfgpu_bloom* b);
/* `jellyfish bc` with the input split between the GPUs (sub_commands/bc_main.cc:84-161, mer_bloom_counter::start :67-71 per
 * rank): every rank creates the same counter (jfgpu_bc_create with the same parameters: same size, same matrices), inserts
 * ITS part of the input, and calls this -- collective.  On return every rank's counter is the counter of the whole input,
 * byte for byte what one counter fed with everything holds (cells saturate at 2 and increments commute,
 * bloom_counter2.hpp:56-107): rank 0 writes the file, `count --bc --gpus` asks it on every rank.  jfgpu_bc_sync then
 * reports the k-mers of all ranks.  _local: the counters of all ranks of a local communicator, in rank order. */
int  jfgpu_comm_bc_merge(jfgpu_comm* c, jfgpu_bloom* b);
int  jfgpu_comm_bc_merge_local(jfgpu_comm* c, jfgpu_bloom** blooms);

/* hash_counter::do_size_doubling(bool) (hash_counter.hpp:78-79).  On (default): the size given at
 * creation is a hint, the table doubles itself (device-side rehash, one more matrix row) before it
 * could exceed 80 % load, as long as device memory allows; jfgpu_get_info / jfgpu_get_matrix report
 * the current geometry.  Off: a full table is the deferred error "Hash full".  The shards of a multi-GPU
 * table grow TOGETHER, inside jfgpu_comm_count_ascii_dev (abi_comm.inl: comm_grow -- every rank draws the next
 * matrix of the same stream, re-shards its entries on the device and the (key, count) pairs travel to their new
 * owners), for keys of one and two words; a shard fed outside a communicator reports "Hash full". */
int  jfgpu_set_growth(jfgpu_table* t, int on);

/* The matrix a table gets when jfgpu_params gives neither matrix_columns nor matrix_seed: the one the
 * reference itself would draw for a table of 2^lsize positions and key_len = 2k bits as the first matrix of
 * its process (RectangularBinaryMatrix::randomize_pseudo_inverse over unseeded glibc random(),
 * lib/rectangular_binary_matrix.cc:240-247, lib/misc.cc:66-72; identity when lsize >= key_len,
 * large_hash_array.hpp:997-1000).  Host only: needs no device.  columns: key_len words, file-header order. */
int  jfgpu_reference_matrix(uint32_t lsize, uint32_t key_len, uint64_t* columns);

/* What jfgpu_count_ascii(_dev) does with every k-mer (mer_counter_base::start, count_main.cc:152-184):
 *   JFGPU_OP_COUNT   add(m, 1)                                   the default
 *   JFGPU_OP_PRIME   set(m): the key enters the table with count 0   first pass of `count --if` (:289-295)
 *   JFGPU_OP_UPDATE  update_add(m, 1): counted only if already there  second pass of `count --if`
 * Keys with count 0 are real entries: stats, histo and dumps report them like the reference does. */
#define JFGPU_OP_COUNT  0
#define JFGPU_OP_PRIME  1
#define JFGPU_OP_UPDATE 2
int  jfgpu_set_operation(jfgpu_table* t, int op);

/* Device memory a table created with (k, size) occupies, and the number of positions it really gets (the size is
 * rounded up to a power of two and to the engine's minimum for that k; large_hash::array::usage_info::mem,
 * sub_commands/mem_main.cc).  Host only.  The partitioned insert path additionally needs its workspace
 * (jfgpu_reserve: about 8 bytes per k-mer fed between two syncs). */
int  jfgpu_table_bytes(uint32_t k, uint64_t size, uint64_t* slots, uint64_t* bytes);

/* Spill instead of "Hash full" when the table may not double (jfgpu_set_growth(t, 0), i.e. `count --disk`):
 * before the table would pass 80 % load the engine applies everything pending and calls fn(user); the callback
 * writes the table out as one sorted run (jfgpu_dump_begin / _next / _end -- what the reference's dumper does
 * from hash_counter::handle_full_ary, hash_counter.hpp:178-198) and returns 0; the engine then empties the table
 * and carries on.  The runs are merged afterwards (jellyfish/merge_files.cc).  fn == NULL: no spilling. */
int  jfgpu_set_spill(jfgpu_table* t, int (*fn)(void* user), void* user);

/* Insert strategy.  0 auto (default), 1 direct (global 64-bit atomics, kernels.hip.hpp),
 * 2 partitioned (radix partition + LDS-resident tiles, kernels_part.hip.hpp; large batches
 * are buffered on the device and applied at the next jfgpu_sync / read).  Results are
 * bit-identical; the environment variable JFGPU_MODE=direct|partitioned sets the default. */
int  jfgpu_set_mode(jfgpu_table* t, int mode);
/* Optional: pre-size the partitioned path's device workspace for `input_bytes` of sequence
 * between two syncs (what `-s` is to the table, hash_counter.hpp:56-64: a hint that moves
 * allocation out of the counting phase).  Without it the workspace grows on demand. */
int  jfgpu_reserve(jfgpu_table* t, uint64_t input_bytes);

/* ---- sequence files parsed on the device -------------------
 * Replaces mer_overlap_sequence_parser::read_fasta / read_fastq
 * (include/jellyfish/mer_overlap_sequence_parser.hpp:160-215): raw FASTA / FASTQ bytes in, the
 * contract buffer (sequence lines concatenated, headers and qualities dropped, one 'N' per record
 * boundary) out -- in device memory, ready for jfgpu_count_ascii_dev / jfgpu_bc_insert_ascii_dev /
 * jfgpu_partition_ascii_dev.
 *
 * A chunk is n <= 2^31 bytes that start at the beginning of a line and end at the end of one.
 *   FASTA  may be cut at ANY line boundary: pass JFGPU_PARSE_CONTINUE for every chunk of a file but
 *          the first and the parser prepends the previous chunk's last k-1 characters (the
 *          reference's "seam", :164-167,182-184), so no k-mer is lost or seen twice.
 *   FASTQ  chunks hold whole 4-line records.  Wrapped sequence / quality lines, blank lines or a
 *          quality string of the wrong length give JFGPU_E_FORMAT and produce nothing: pass that
 *          chunk to the host parser, which implements the reference's general reader and its
 *          "Invalid fastq sequence" error (:292-309).
 * The returned buffer belongs to the parser and stays valid until the second-next parse call
 * (two buffers alternate), so one chunk can be counted while the next is parsed.  Calls are
 * synchronous: on return the buffer is complete. */
typedef struct jfgpu_parser jfgpu_parser;
#define JFGPU_PARSE_FASTA    1u
#define JFGPU_PARSE_FASTQ    2u
#define JFGPU_PARSE_CONTINUE 4u
int  jfgpu_parser_create(int device, uint32_t k, jfgpu_parser** out);
void jfgpu_parser_destroy(jfgpu_parser* p);
/* d_bytes: device memory, readable up to the next multiple of 16 bytes. */
int  jfgpu_parser_parse_dev(jfgpu_parser* p, const char* d_bytes, size_t n, unsigned flags,
                            const char** d_out, size_t* n_out, uint64_t* n_records);
/* bytes: host memory (e.g. the mmap of the file); copied to the device, then as above. */
int  jfgpu_parser_parse(jfgpu_parser* p, const char* bytes, size_t n, unsigned flags,
                        const char** d_out, size_t* n_out, uint64_t* n_records);
/* Pinned host staging owned by the parser (which = 0 or 1, grown on demand): fill one while the other
 * is being parsed, then pass it to jfgpu_parser_parse -- the host-to-device copy of a pinned buffer runs
 * at PCIe speed, that of pageable memory (an mmap'ed file) at a fraction of it. */
int  jfgpu_parser_host_buffer(jfgpu_parser* p, int which, size_t bytes, char** out);
/* Milliseconds the parse kernels of the last call took on the device (HIP events). */
/* The same in pipeline form: _upload(which) starts the copy of a (pinned) host buffer to device buffer `which` on a copy
 * stream and returns; _upload_wait(which) says when that host buffer may be refilled; _parse_uploaded(which) parses what
 * was uploaded (waiting for that copy only).  Two buffers each side: the copy of chunk i+1 overlaps parse and count of
 * chunk i while the host reads chunk i+2. */
int  jfgpu_parser_upload(jfgpu_parser* p, int which, const char* bytes, size_t n);
int  jfgpu_parser_upload_wait(jfgpu_parser* p, int which);
int  jfgpu_parser_parse_uploaded(jfgpu_parser* p, int which, unsigned flags,
                                 const char** d_out, size_t* n_out, uint64_t* n_records);
/* count -Q / --min-quality (mer_qual_iterator.hpp:75-84, count_main.cc:234-256): from now on a FASTQ base whose quality
 * character is below min_qual_char reaches the contract buffer as 'N'.  0 switches it off.  The raw chunk given to
 * jfgpu_parser_parse_dev is edited in place when this is on.  FASTA input has no qualities: unaffected, as in the reference. */
int  jfgpu_parser_set_min_quality(jfgpu_parser* p, int min_qual_char);
int  jfgpu_parser_last_ms(jfgpu_parser* p, double* ms);

/* ---- measurement helpers (bench.py; not part of the reference surface) -- */
/* Per-kernel HIP-event timing on the table's stream.  which: 0 count (direct), 1 add_keys,
 * 2 shard partition, 3 lookup, 4 P1 partition, 5 P2 partition, 6 tile insert, 7 items-direct.  Returns accumulated milliseconds and launches since the
 * last reset. */
int  jfgpu_profile_enable(jfgpu_table* t, int on);
int  jfgpu_profile_get(jfgpu_table* t, int which, double* ms, uint64_t* launches, uint64_t* units);
int  jfgpu_profile_reset(jfgpu_table* t);
/* The same spans one by one, in launch order, since the last reset (at most 65536 are kept): which[i] is the slot above,
 * ms[i] the span's device time.  *n = spans recorded (may exceed cap: only cap are written).  bench.py --gpus N rebuilds
 * every step's route / split / partition / insert times from it, per rank. */
int  jfgpu_profile_spans(jfgpu_table* t, int* which, double* ms, size_t cap, size_t* n);
/* The engine's event counters since the last jfgpu_clear, after waiting for the table's stream (tests assert from them
 * which code path a flush took; `jellyfish-amd count --timing` reports them).  out[0..n): 0 tile-full events ("Hash
 * full" is raised from it), 1 k-mer occurrences fed, 2 overflow side table full, 3 side-table entries, 4 k-mers sent to
 * a shard that does not own them, 5 items inserted with global atomics by the partition kernels (ring / region
 * overflow, runs of one k-mer), 6 / 7 items placed / items past rank 3 in the sampling launch of the last flush that
 * sampled (host_partition.inl), 8 flushes that ran the plain tile kernel, 9 flushes that ran the HEAVY one, 10-13 launches
 * of the second partition level by kernel (loader / storer rings, shared rings, sort-based single pass, exact count +
 * scatter), 14 / 15 launches of the first level (ring kernel / any other). */
#define JFGPU_N_COUNTERS 16
int  jfgpu_get_counters(jfgpu_table* t, uint64_t* out, uint32_t n);
/* Synthetic reads: n_reads records of read_len uniform iid bases, each followed by
 * one 'N' separator (the contract buffer the parser would produce for a FASTA of
 * such reads); counter-based RNG so any slice is reproducible.  d_out needs
 * n_reads * (read_len + 1) bytes.  first_read offsets the read index. */
int  jfgpu_gen_reads_dev(jfgpu_table* t, char* d_out, uint64_t first_read, uint64_t n_reads, uint32_t read_len, uint64_t seed);
/* Synthetic reads of the secondary distribution "G" (BASELINE.md section 3): read_len-base windows at uniform
 * positions and strands of a uniform random genome of genome_len bases (a pure function of the seed, never stored),
 * each base substituted with probability substitution_rate; same layout and slicing rules as jfgpu_gen_reads_dev.
 * High coverage means most k-mers repeat: the case where duplicates aggregate in the LDS tiles. */
int  jfgpu_gen_genome_reads_dev(jfgpu_table* t, char* d_out, uint64_t first_read, uint64_t n_reads, uint32_t read_len,
                                uint64_t genome_len, double substitution_rate, uint64_t seed);
/* Random-access roofline denominator: n independent 64-bit atomicAdds
 * at uniformly random slots of this table's own memory (table must be cleared
 * afterwards).  Returns updates per second. */
int  jfgpu_gups(jfgpu_table* t, uint64_t n_updates, int mode, double* updates_per_s);
/* Pinned host memory: buffers handed to jfgpu_dump_next / jfgpu_count_ascii / jfgpu_memcpy_* move at PCIe speed
 * when they come from here (pageable memory goes through the runtime's staging at a fraction of it). */
int  jfgpu_malloc_host(size_t bytes, void** out);
int  jfgpu_free_host(void* p);
/* Raw device memory for callers without torch. */
int  jfgpu_malloc_dev(jfgpu_table* t, size_t bytes, void** out);
int  jfgpu_free_dev(jfgpu_table* t, void* p);
int  jfgpu_memcpy_h2d(jfgpu_table* t, void* d_dst, const void* src, size_t bytes);
int  jfgpu_memcpy_d2h(jfgpu_table* t, void* dst, const void* d_src, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* JFGPU_H */
