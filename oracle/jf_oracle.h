/* oracle/jf_oracle.h -- TEST INFRASTRUCTURE ONLY (parity oracle).
 *
 * Plain-C restatement of the reference's `jellyfish count` hot path
 * (feed -> 2-bit rolling encode (+canonical) -> GF(2) hash; exact counts by
 * sort+run-length instead of the lock-free table, whose observable result is
 * the same {k-mer -> count} map).  Every function cites the reference
 * file:line (relative to /root/reference) it follows.
 *
 * PARITY PINNED: tests/test_oracle.py checks this restatement against
 *   (1) oracle/_ref (the reference's own classes compiled in place), whose
 *       outputs reproduce the reference's golden md5s
 *       (tests/parallel_hashing.sh:7-19), and
 *   (2) the committed fixtures under tests/golden/ generated from oracle/_ref
 *       by oracle/gen_golden.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * this library; the product (libjfgpu.so) never links or calls it.
 */
#ifndef JF_ORACLE_H
#define JF_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define JFO_MAX_WORDS 8 /* k <= 256 */

/* include/jellyfish/mer_dna.hpp:38-55 : A a->0 C c->1 G g->2 T t->3, everything else < 0
 * (-1 IUPAC/'-', -2 '\n', -3 other) */
int jfo_code(unsigned char c);

/* Number of 64-bit words of a k-mer: include/jellyfish/mer_dna.hpp:160-170 (nb_words) */
unsigned jfo_nb_words(unsigned k);

/* Net effect of mer_overlap_sequence_parser on ONE file
 * (include/jellyfish/mer_overlap_sequence_parser.hpp:120-217,260-307): strips
 * headers / newlines / CR / quality lines, concatenates the sequence lines of a
 * record, writes one 'N' between records.  Seams are an implementation detail of
 * the 4 KiB buffering and do not change the k-mer multiset, so the whole file
 * becomes one contract buffer.  Returns the number of bytes written to out
 * (out_cap >= n is always enough) or (size_t)-1 on "Unsupported format" /
 * "Invalid fastq sequence". */
size_t jfo_parse_file(const char* data, size_t n, char* out, size_t out_cap);

/* mer_iterator (include/jellyfish/mer_iterator.hpp:51,53-81) over one contract
 * buffer: every window of k valid bases, optionally canonical (numerically
 * smaller of the mer and its reverse complement, mer_dna.hpp:227-250,401-431).
 * Writes nb_words(k) words per k-mer (word 0 least significant) into out
 * (capacity cap k-mers); returns the number of k-mers found (may exceed cap:
 * only the first cap are written). */
size_t jfo_extract(const char* seq, size_t n, unsigned k, int canonical, uint64_t* out, size_t cap);

/* Reverse complement / canonical of one k-mer in place (mer_dna.hpp:376-431). */
void jfo_revcomp(const uint64_t* in, uint64_t* out, unsigned k);
int  jfo_less(const uint64_t* a, const uint64_t* b, unsigned nw); /* mer_dna.hpp:227-250 */

/* Exact counting: sorts the nw-word k-mers in place (numeric order) and
 * run-length encodes them into keys[] (nw words each) / counts[].  Returns the
 * number of distinct k-mers.  keys/counts need capacity n. */
size_t jfo_sort_count(uint64_t* kmers, size_t n, unsigned nw, uint64_t* keys, uint64_t* counts);

/* RectangularBinaryMatrix::times (include/jellyfish/rectangular_binary_matrix.hpp:223-261):
 * columns[] in file-header order (c = 2k entries of r bits); key bit j (LSB = 0)
 * selects columns[c-1-j].  columns == NULL => identity: key[0] & (2^r - 1). */
uint64_t jfo_matrix_times(const uint64_t* columns, unsigned r, unsigned c, const uint64_t* key);

/* k-mer <-> string (mer_dna.hpp:434-446,526-542): base 0 of the string is the most significant. */
void jfo_to_str(const uint64_t* key, unsigned k, char* out /* k+1 */);
int  jfo_from_str(const char* s, unsigned k, uint64_t* key);

/* Bloom counter of `jellyfish bc` (config 3):
 * include/jellyfish/bloom_counter2.hpp:56-107 (insert), :109-142 (check),
 * bloom_common.hpp:61-79.  data = ceil(m/5) bytes, 5 base-3 digits per byte.
 * insert returns the minimum previous digit (0,1,2). */
unsigned jfo_bc_insert(uint8_t* data, uint64_t m, unsigned nb_hashes, uint64_t h0, uint64_t h1);
unsigned jfo_bc_check(const uint8_t* data, uint64_t m, unsigned nb_hashes, uint64_t h0, uint64_t h1);

#ifdef __cplusplus
}
#endif
#endif
