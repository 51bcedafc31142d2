/* oracle/jf_oracle.c -- TEST INFRASTRUCTURE ONLY (see jf_oracle.h).
 * Plain-C restatement of the reference hot path; citations are relative to
 * /root/reference.  Never linked into the product library. */
#include "jf_oracle.h"
#include <stdlib.h>
#include <string.h>

/* include/jellyfish/mer_dna.hpp:38-55 */
int jfo_code(unsigned char c) {
  switch(c) {
  case 'A': case 'a': return 0;
  case 'C': case 'c': return 1;
  case 'G': case 'g': return 2;
  case 'T': case 't': return 3;
  case '\n': return -2;
  case '-':
  case 'B': case 'D': case 'H': case 'K': case 'M': case 'N': case 'R':
  case 'S': case 'V': case 'W': case 'X': case 'Y':
  case 'b': case 'd': case 'h': case 'k': case 'm': case 'n': case 'r':
  case 's': case 'v': case 'w': case 'x': case 'y':
    return -1;
  default: return -3;
  }
}

unsigned jfo_nb_words(unsigned k) { return (2 * k + 63) / 64; }

/* ---- feed: mer_overlap_sequence_parser.hpp ------------------------------ */
typedef struct { const char* p; const char* end; } cursor;
static int  cpeek(const cursor* c) { return c->p < c->end ? (unsigned char)*c->p : -1; }
static void ignore_line(cursor* c) {                       /* :276-278 */
  while(c->p < c->end && *c->p != '\n') ++c->p;
  if(c->p < c->end) ++c->p;
}
static int skip_newlines(cursor* c) {                      /* :280-287 */
  int dos = 0;
  while(cpeek(c) == '\n' || cpeek(c) == '\r') { dos |= (*c->p == '\r'); ++c->p; }
  return dos;
}
/* read_sequence :260-274 -- copy whole lines until a line starts with `stop`;
 * trailing CRs of each line are dropped. */
static size_t read_sequence(cursor* c, char* out, char stop, char* le) {
  size_t n = 0;
  *le = '\n';
  skip_newlines(c);
  while(c->p < c->end && cpeek(c) != stop) {
    while(c->p < c->end && *c->p != '\n') out[n++] = *c->p++;
    while(n > 0 && out[n - 1] == '\r') { *le = '\r'; --n; }
    if(skip_newlines(c)) *le = '\r';
  }
  return n;
}
/* skip_quals :290-307 ; returns 0 on "Invalid fastq sequence" */
static int skip_quals(cursor* c, size_t read_len, char le) {
  ignore_line(c);
  size_t quals = 0;
  skip_newlines(c);
  while(c->p < c->end && quals < read_len) {
    /* istream::ignore(read_len - quals + 1, le): extract up to that many chars, stop after le */
    size_t lim = read_len - quals + 1, got = 0;
    int hit_eof = 0;
    while(got < lim) {
      if(c->p >= c->end) { hit_eof = 1; break; }
      char ch = *c->p++; ++got;
      if(ch == le) break;
    }
    quals += got;
    if(!hit_eof) ++read_len;
    skip_newlines(c);
  }
  skip_newlines(c);
  if(quals == read_len && (cpeek(c) == '@' || cpeek(c) == -1)) return 1;
  return 0;
}

size_t jfo_parse_file(const char* data, size_t n, char* out, size_t out_cap) {
  (void)out_cap;
  cursor c = { data, data + n };
  size_t w = 0;
  if(n == 0) return 0;                                     /* :135 empty file skipped */
  if(data[0] == '>') {                                     /* :136-140 FASTA */
    ignore_line(&c);
    while(c.p < c.end) {                                   /* read_fasta :161-185 */
      char le;
      w += read_sequence(&c, out + w, '>', &le);
      if(cpeek(&c) == '>') {
        if(w > 0) out[w++] = 'N';
        ignore_line(&c);
      }
    }
    return w;
  }
  if(data[0] == '@') {                                     /* :141-145 FASTQ */
    ignore_line(&c);
    size_t seq_len = 0;
    while(c.p < c.end) {                                   /* read_fastq :187-217 */
      char le;
      size_t nread = read_sequence(&c, out + w, '+', &le);
      w += nread; seq_len += nread;
      if(cpeek(&c) == '+') {
        if(!skip_quals(&c, seq_len, le)) return (size_t)-1;
        if(c.p < c.end) {
          out[w++] = 'N';
          ignore_line(&c);
        }
        seq_len = 0;
      }
    }
    return w;
  }
  return (size_t)-1;                                       /* :146-147 Unsupported format */
}

/* ---- encode: mer_dna.hpp shift_left :322-345, shift_right :347-370 ------ */
static void shift_left(uint64_t* m, unsigned k, unsigned nw, unsigned code) {
  for(unsigned i = nw; i-- > 1; ) m[i] = (m[i] << 2) | (m[i - 1] >> 62);
  m[0] = (m[0] << 2) | code;
  unsigned top = (2 * k) & 63;
  if(top) m[nw - 1] &= (~(uint64_t)0) >> (64 - top);
}
static void shift_right(uint64_t* m, unsigned k, unsigned nw, unsigned code) {
  for(unsigned i = 0; i + 1 < nw; ++i) m[i] = (m[i] >> 2) | (m[i + 1] << 62);
  unsigned topbits = 2 * k - 64 * (nw - 1);                /* bits in top word */
  m[nw - 1] = (m[nw - 1] >> 2) | ((uint64_t)code << (topbits - 2));
}
int jfo_less(const uint64_t* a, const uint64_t* b, unsigned nw) {
  for(unsigned i = nw; i-- > 0; ) if(a[i] != b[i]) return a[i] < b[i];
  return 0;
}

size_t jfo_extract(const char* seq, size_t n, unsigned k, int canonical, uint64_t* out, size_t cap) {
  unsigned nw = jfo_nb_words(k), filled = 0;
  uint64_t m[JFO_MAX_WORDS] = {0}, rc[JFO_MAX_WORDS] = {0};
  size_t   found = 0;
  for(size_t i = 0; i < n; ++i) {                          /* mer_iterator.hpp:67-76 */
    int code = jfo_code((unsigned char)seq[i]);
    if(code >= 0) {
      shift_left(m, k, nw, (unsigned)code);
      shift_right(rc, k, nw, 3u - (unsigned)code);
      if(filled < k) ++filled;
    } else filled = 0;
    if(filled >= k) {
      if(found < cap) {
        const uint64_t* src = (!canonical || jfo_less(m, rc, nw)) ? m : rc;   /* :51 */
        memcpy(out + found * nw, src, nw * sizeof(uint64_t));
      }
      ++found;
    }
  }
  return found;
}

void jfo_revcomp(const uint64_t* in, uint64_t* out, unsigned k) {
  unsigned nw = jfo_nb_words(k);
  uint64_t tmp[JFO_MAX_WORDS] = {0};
  for(unsigned i = 0; i < k; ++i) {                        /* base i from the right end */
    unsigned code = (unsigned)((in[i / 32] >> (2 * (i % 32))) & 3);
    unsigned j = k - 1 - i;
    tmp[j / 32] |= (uint64_t)(3 - code) << (2 * (j % 32));
  }
  memcpy(out, tmp, nw * sizeof(uint64_t));
}

/* ---- exact counts ------------------------------------------------------- */
static unsigned g_nw;
static int cmp_words(const void* a, const void* b) {
  const uint64_t* x = (const uint64_t*)a; const uint64_t* y = (const uint64_t*)b;
  for(unsigned i = g_nw; i-- > 0; ) if(x[i] != y[i]) return x[i] < y[i] ? -1 : 1;
  return 0;
}
size_t jfo_sort_count(uint64_t* kmers, size_t n, unsigned nw, uint64_t* keys, uint64_t* counts) {
  if(n == 0) return 0;
  g_nw = nw;
  qsort(kmers, n, nw * sizeof(uint64_t), cmp_words);
  size_t d = 0;
  memcpy(keys, kmers, nw * sizeof(uint64_t)); counts[0] = 1;
  for(size_t i = 1; i < n; ++i) {
    if(memcmp(kmers + i * nw, keys + d * nw, nw * sizeof(uint64_t)) == 0) ++counts[d];
    else { ++d; memcpy(keys + d * nw, kmers + i * nw, nw * sizeof(uint64_t)); counts[d] = 1; }
  }
  return d + 1;
}

/* ---- hash: rectangular_binary_matrix.hpp:223-261 ------------------------ */
uint64_t jfo_matrix_times(const uint64_t* columns, unsigned r, unsigned c, const uint64_t* key) {
  if(!columns) return key[0] & (r >= 64 ? ~(uint64_t)0 : (((uint64_t)1 << r) - 1));
  uint64_t res = 0;
  for(unsigned j = 0; j < c; ++j)
    if((key[j / 64] >> (j % 64)) & 1) res ^= columns[c - 1 - j];
  return res;
}

/* ---- strings: mer_dna.hpp:434-446, 526-542 ------------------------------ */
void jfo_to_str(const uint64_t* key, unsigned k, char* out) {
  static const char rev[4] = { 'A', 'C', 'G', 'T' };
  for(unsigned i = 0; i < k; ++i) {
    unsigned j = k - 1 - i;                                /* leftmost char = most significant */
    out[i] = rev[(key[j / 32] >> (2 * (j % 32))) & 3];
  }
  out[k] = 0;
}
int jfo_from_str(const char* s, unsigned k, uint64_t* key) {
  unsigned nw = jfo_nb_words(k);
  memset(key, 0, nw * sizeof(uint64_t));
  for(unsigned i = 0; i < k; ++i) {
    int code = jfo_code((unsigned char)s[i]);
    if(code < 0) return 0;
    unsigned j = k - 1 - i;
    key[j / 32] |= (uint64_t)code << (2 * (j % 32));
  }
  return 1;
}

/* ---- Bloom counter: bloom_counter2.hpp:56-142 --------------------------- */
static const unsigned pow3[5] = { 1, 3, 9, 27, 81 };
unsigned jfo_bc_insert(uint8_t* data, uint64_t m, unsigned nb_hashes, uint64_t h0, uint64_t h1) {
  const uint64_t base = h0 % m, inc = h1 % m;
  unsigned res = 2;
  for(unsigned i = 0; i < nb_hashes; ++i) {
    uint64_t p = (base + (uint64_t)i * inc) % m;
    uint8_t* b = data + p / 5;
    unsigned  w = (*b / pow3[p % 5]) % 3;
    if(w == 2) continue;
    *b = (uint8_t)(*b + pow3[p % 5]);
    if(w < res) res = w;
  }
  return res;
}
unsigned jfo_bc_check(const uint8_t* data, uint64_t m, unsigned nb_hashes, uint64_t h0, uint64_t h1) {
  const uint64_t base = h0 % m, inc = h1 % m;
  unsigned res = 2;
  for(unsigned i = 0; i < nb_hashes; ++i) {
    uint64_t p = (base + (uint64_t)i * inc) % m;
    unsigned  w = (data[p / 5] / pow3[p % 5]) % 3;
    if(w < res) res = w;
  }
  return res;
}
