// oracle/ref_drivers/ref_jf.cc -- TEST INFRASTRUCTURE, not product code.
//
// A thin driver around the *reference's own classes* (headers + lib/*.cc compiled
// in place from /root/reference by oracle/Makefile; nothing is copied).  It
// exists because the reference CLI cannot be built here: every sub-command
// includes a yaggo-generated *_cmdline.hpp and yaggo is absent.  The driver
// wires the classes together the same way the reference mains do:
//   count : sub_commands/count_main.cc:142-184,275-284,331-355
//   dump  : sub_commands/dump_main.cc:36-88
//   histo : sub_commands/histo_main.cc:34-90
//   stats : sub_commands/stats_main.cc:33-79
//   query : sub_commands/query_main.cc:44-123
//   bc    : sub_commands/bc_main.cc:51-72,109-148 ; count --bc : count_main.cc:99-131,191-206,313-316
// so its outputs ARE the reference's outputs for the hot path.  Only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may execute it.
#include <config.h>

#include <cstdlib>
#include <cstring>
#include <chrono>
#include <fstream>
#include <iostream>
#include <limits>
#include <memory>
#include <string>
#include <vector>

#include <jellyfish/err.hpp>
#include <jellyfish/thread_exec.hpp>
#include <jellyfish/hash_counter.hpp>
#include <jellyfish/stream_manager.hpp>
#include <jellyfish/mer_overlap_sequence_parser.hpp>
#include <jellyfish/mer_iterator.hpp>
#include <jellyfish/whole_sequence_parser.hpp>
#include <jellyfish/mer_qual_iterator.hpp>
#include <jellyfish/mapped_file.hpp>
#include <jellyfish/mer_dna_bloom_counter.hpp>
#include <jellyfish/jellyfish.hpp>

using jellyfish::mer_dna;
typedef std::vector<const char*>                                         file_vector;
typedef jellyfish::stream_manager<file_vector::const_iterator>           stream_manager_type;
typedef jellyfish::mer_overlap_sequence_parser<stream_manager_type>      sequence_parser;
typedef jellyfish::mer_iterator<sequence_parser, mer_dna>                mer_iterator_type;

typedef jellyfish::whole_sequence_parser<stream_manager_type>            read_parser;
typedef jellyfish::mer_qual_iterator<read_parser, mer_dna>               mer_qual_iterator_type;

static double now_s() {
  using namespace std::chrono;
  return duration_cast<duration<double>>(steady_clock::now().time_since_epoch()).count();
}

// Same loop bodies as mer_counter_base::start (count_main.cc:152-184): COUNT, and the PRIME / UPDATE pair
// of `count --if` (:289-295).
enum ref_operation { REF_COUNT, REF_PRIME, REF_UPDATE };
class ref_counter : public jellyfish::thread_exec {
  mer_hash&       ary_;
  sequence_parser parser_;
  bool            canonical_;
  const jellyfish::mer_dna_bloom_counter* bc_;   // count --bc filter (count_main.cc:109-119), may be null
  ref_operation   op_;
public:
  std::vector<size_t> counts_;
  ref_counter(int nb_threads, mer_hash& ary, stream_manager_type& streams, bool canonical,
              const jellyfish::mer_dna_bloom_counter* bc = 0, ref_operation op = REF_COUNT)
    : ary_(ary), parser_(mer_dna::k(), streams.nb_streams(), 3 * nb_threads, 4096, streams),
      canonical_(canonical), bc_(bc), op_(op), counts_(nb_threads, 0) { ary_.reset_done(); }
  virtual void start(int thid) {
    size_t count = 0;
    mer_dna tmp;
    for(mer_iterator_type mers(parser_, canonical_); mers; ++mers) {
      if(!bc_ || bc_->check(*mers) > 1) {
        if(op_ == REF_COUNT) ary_.add(*mers, 1);
        else if(op_ == REF_PRIME) ary_.set(*mers);
        else ary_.update_add(*mers, 1, tmp);
      }
      ++count;
    }
    counts_[thid] = count;
    ary_.done();
  }
};

// ---- content digest: the checksum include/jfgpu.h documents for jfgpu_digest, computed from the REFERENCE's table
// (large_hash_array's own iterators) or from a reference-readable file.  { records, sum counts, sum h, xor h }.
static inline uint64_t digest_mix(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
struct digest_t {
  uint64_t n, total, sum, x;
  digest_t() : n(0), total(0), sum(0), x(0) { }
  void add(const mer_dna& m, uint64_t val) {
    uint64_t h = 0x9E3779B97F4A7C15ull;
    const unsigned words = (mer_dna::k() + 31) / 32;
    for(unsigned w = 0; w < words; ++w) h = digest_mix(h ^ m.word(w));
    h = digest_mix(h ^ val);
    ++n; total += val; sum += h; x ^= h;
  }
  void merge(const digest_t& o) { n += o.n; total += o.total; sum += o.sum; x ^= o.x; }
  void print(std::ostream& os) const { os << "records " << n << "\ntotal " << total << "\nsum " << sum << "\nxor " << x << "\n"; }
};
class ref_digester : public jellyfish::thread_exec {
  mer_array* ary_; int nb_; uint64_t lower_, upper_;
public:
  std::vector<digest_t> parts_;
  ref_digester(mer_array* ary, int nb_threads, uint64_t lower, uint64_t upper) : ary_(ary), nb_(nb_threads), lower_(lower), upper_(upper), parts_(nb_threads) { }
  virtual void start(int thid) {
    digest_t d;
    mer_array::eager_iterator it = ary_->eager_slice(thid, nb_);
    while(it.next()) if(it.val() >= lower_ && it.val() <= upper_) d.add(it.key(), it.val());
    parts_[thid] = d;
  }
};

// count -Q / --min-quality: the quality-aware loop of count_main.cc:86-92,326-329 (whole_sequence_parser + mer_qual_iterator)
class ref_qual_counter : public jellyfish::thread_exec {
  mer_hash&   ary_;
  read_parser parser_;
  bool        canonical_;
  char        min_qual_;
public:
  std::vector<size_t> counts_;
  ref_qual_counter(int nb_threads, mer_hash& ary, stream_manager_type& streams, bool canonical, char min_qual)
    : ary_(ary), parser_(4 * nb_threads, 100, streams.nb_streams(), streams), canonical_(canonical), min_qual_(min_qual), counts_(nb_threads, 0) { ary_.reset_done(); }
  virtual void start(int thid) {
    size_t count = 0;
    for(mer_qual_iterator_type mers(parser_, min_qual_, canonical_); mers; ++mers) { ary_.add(*mers, 1); ++count; }
    counts_[thid] = count;
    ary_.done();
  }
};

static uint64_t parse_size(const char* s) {  // yaggo "suffix": k/M/G/T = powers of 1000
  char* end;
  double v = strtod(s, &end);
  switch(*end) {
  case 'k': v *= 1e3; break;  case 'M': v *= 1e6; break;
  case 'G': v *= 1e9; break;  case 'T': v *= 1e12; break;
  default: break;
  }
  return (uint64_t)v;
}

static int do_count(int argc, char* argv[]) {
  unsigned    k = 0, threads = 1, counter_len = 7, reprobes = 126, out_counter_len = 4, Files = 1;
  uint64_t    size = 0, lower = 0, upper = std::numeric_limits<uint64_t>::max();
  bool        canonical = false, text = false, no_write = false, lower_given = false, upper_given = false;
  const char* output = "mer_counts.jf";
  const char* timing = 0;
  const char* bc_path = 0;
  const char* digest_path = 0;
  int min_qual = 0;
  file_vector files, if_files;
  for(int i = 1; i < argc; ++i) {
    std::string a(argv[i]);
    auto next = [&]() -> const char* { if(i + 1 >= argc) { std::cerr << "missing value for " << a << "\n"; exit(1); } return argv[++i]; };
    if(a == "-m") k = atoi(next());
    else if(a == "--if") if_files.push_back(next());
    else if(a == "-s") size = parse_size(next());
    else if(a == "-t") threads = atoi(next());
    else if(a == "-c") counter_len = atoi(next());
    else if(a == "-p") reprobes = atoi(next());
    else if(a == "-F") Files = atoi(next());
    else if(a == "--out-counter-len") out_counter_len = atoi(next());
    else if(a == "-o") output = next();
    else if(a == "-L") { lower = strtoull(next(), 0, 10); lower_given = true; }
    else if(a == "-U") { upper = strtoull(next(), 0, 10); upper_given = true; }
    else if(a == "-C") canonical = true;
    else if(a == "--text") text = true;
    else if(a == "--no-write") no_write = true;
    else if(a == "--timing") timing = next();
    else if(a == "--bc") bc_path = next();
    else if(a == "--digest") digest_path = next();
    else if(a == "-Q") min_qual = next()[0];
    else files.push_back(argv[i]);
  }
  if(!k || !size || files.empty()) {
    std::cerr << "usage: ref_jf count -m K -s SIZE [-t T] [-C] [-c bits] [-p reprobes] [--out-counter-len B] [-o out] [-L l] [-U u] [--text] [--timing f] [--no-write] files...\n";
    return 1;
  }
  double t0 = now_s();
  jellyfish::file_header header;
  header.fill_standard();
  header.set_cmdline(argc, argv);
  mer_dna::k(k);
  header.canonical(canonical);
  mer_hash ary(size, k * 2, counter_len, threads, reprobes);
  std::unique_ptr<jellyfish::dumper_t<mer_array> > dumper;
  if(text) dumper.reset(new text_dumper(threads, output, &header));
  else     dumper.reset(new binary_dumper(out_counter_len, ary.key_len(), threads, output, &header));
  ary.dumper(dumper.get());
  double t1 = now_s();

  std::unique_ptr<jellyfish::mer_dna_bloom_counter> bc;
  if(bc_path) {   // load_bloom_filter, count_main.cc:191-206
    std::ifstream in(bc_path, std::ios::in | std::ios::binary);
    jellyfish::file_header bh(in);
    if(!in.good() || bh.format() != "bloomcounter" || bh.key_len() != k * 2) { std::cerr << "bad bloom counter file\n"; return 1; }
    jellyfish::hash_pair<mer_dna> fns(bh.matrix(1), bh.matrix(2));
    bc.reset(new jellyfish::mer_dna_bloom_counter(bh.size(), bh.nb_hashes(), in, fns));
  }
  ref_operation op = REF_COUNT;
  if(!if_files.empty()) {   // count_main.cc:289-295: prime the hash with the mers of the --if files, then only update
    stream_manager_type if_streams(Files);
    if_streams.paths(if_files.begin(), if_files.end());
    ref_counter primer(threads, ary, if_streams, canonical, 0, REF_PRIME);
    primer.exec_join(threads);
    op = REF_UPDATE;
  }
  stream_manager_type streams(Files);
  streams.paths(files.begin(), files.end());
  size_t total = 0;
  if(min_qual) {
    ref_qual_counter counter(threads, ary, streams, canonical, (char)min_qual);
    counter.exec_join(threads);
    for(size_t c : counter.counts_) total += c;
  } else {
    ref_counter counter(threads, ary, streams, canonical, bc.get(), op);
    counter.exec_join(threads);
    for(size_t c : counter.counts_) total += c;
  }
  double t2 = now_s();

  if(digest_path) {     // before the dump: the sorted dumper zeroes the table behind itself
    ref_digester dg(ary.ary(), threads, lower_given ? lower : 0, upper_given ? upper : std::numeric_limits<uint64_t>::max());
    dg.exec_join(threads);
    digest_t d;
    for(size_t i = 0; i < dg.parts_.size(); ++i) d.merge(dg.parts_[i]);
    std::ofstream df(digest_path);
    d.print(df);
  }
  if(!no_write) {
    dumper->one_file(true);
    if(lower_given) dumper->min(lower);
    if(upper_given) dumper->max(upper);
    dumper->dump(ary.ary());
  }
  double t3 = now_s();
  if(timing) {
    std::ofstream tf(timing);
    tf << "Init     " << (t1 - t0) << "\n"
       << "Counting " << (t2 - t1) << "\n"
       << "Writing  " << (t3 - t2) << "\n"
       << "Mers     " << total << "\n";
  }
  return 0;
}


// ---- bc: jellyfish bc (Bloom counter first pass), bc_main.cc:51-72,109-148 ----
class ref_bloom_counter : public jellyfish::thread_exec {
  jellyfish::mer_dna_bloom_counter& filter_;
  sequence_parser                   parser_;
  bool                              canonical_;
public:
  ref_bloom_counter(int nb_threads, jellyfish::mer_dna_bloom_counter& filter, stream_manager_type& streams, bool canonical)
    : filter_(filter), parser_(mer_dna::k(), streams.nb_streams(), 3 * nb_threads, 4096, streams), canonical_(canonical) { }
  virtual void start(int thid) {
    for(mer_iterator_type mers(parser_, canonical_); mers; ++mers) filter_.insert(*mers);
  }
};

static int do_bc(int argc, char* argv[]) {
  unsigned k = 0, threads = 1; uint64_t size = 0; double fpr = 0.001; bool canonical = false;
  const char* output = "mer_bloom_filter";
  file_vector files;
  for(int i = 1; i < argc; ++i) {
    std::string a(argv[i]);
    if(a == "-m") k = atoi(argv[++i]);
    else if(a == "-s") size = parse_size(argv[++i]);
    else if(a == "-t") threads = atoi(argv[++i]);
    else if(a == "-f") fpr = atof(argv[++i]);
    else if(a == "-o") output = argv[++i];
    else if(a == "-C") canonical = true;
    else files.push_back(argv[i]);
  }
  if(!k || !size || files.empty()) { std::cerr << "usage: ref_jf bc -m K -s N [-f fpr] [-C] [-t T] [-o out] files...\n"; return 1; }
  jellyfish::file_header header;
  header.fill_standard();
  header.set_cmdline(argc, argv);
  mer_dna::k(k);
  header.canonical(canonical);
  std::ofstream out(output);
  if(!out.good()) { std::cerr << "Can't open output file\n"; return 1; }
  header.format("bloomcounter");
  header.key_len(k * 2);
  jellyfish::hash_pair<mer_dna> hash_fns;
  header.matrix(hash_fns.m1, 1);
  header.matrix(hash_fns.m2, 2);
  jellyfish::mer_dna_bloom_counter filter(fpr, size, hash_fns);
  header.size(filter.m());
  header.nb_hashes(filter.k());
  header.write(out);
  stream_manager_type streams(1);
  streams.paths(files.begin(), files.end());
  ref_bloom_counter counter(threads, filter, streams, canonical);
  counter.exec_join(threads);
  filter.write_bits(out);
  out.close();
  return 0;
}

static bool open_db(const char* path, std::ifstream& is, jellyfish::file_header& header) {
  is.open(path);
  if(!is.good()) { std::cerr << "Failed to open '" << path << "'\n"; return false; }
  if(!header.read(is)) { std::cerr << "Failed to parse header of '" << path << "'\n"; return false; }
  mer_dna::k(header.key_len() / 2);
  return true;
}

template<typename Reader>
static void dump_loop(Reader& r, bool column, char spacer, uint64_t lower, uint64_t upper) {
  while(r.next()) {
    if(r.val() < lower || r.val() > upper) continue;
    if(column) std::cout << r.key() << spacer << r.val() << "\n";
    else       std::cout << ">" << r.val() << "\n" << r.key() << "\n";
  }
}

// dump [-c] [-t] [-L l] [-U u] db   (dump_main.cc:36-52)
static int do_dump(int argc, char* argv[]) {
  bool column = false, tab = false, check_order = false;
  uint64_t lower = 0, upper = std::numeric_limits<uint64_t>::max();
  const char* db = 0;
  for(int i = 1; i < argc; ++i) {
    std::string a(argv[i]);
    if(a == "-c") column = true; else if(a == "-t") tab = true;
    else if(a == "--check-order") check_order = true;
    else if(a == "-L") lower = strtoull(argv[++i], 0, 10);
    else if(a == "-U") upper = strtoull(argv[++i], 0, 10);
    else db = argv[i];
  }
  if(!db) return 1;
  std::ios::sync_with_stdio(false);
  if(check_order) {
    // Structural validity as the reference's own readers need it: strictly
    // ascending (pos, key) under the header's matrix (mer_heap.hpp:26-30).
    std::ifstream is; jellyfish::file_header header;
    if(!open_db(db, is, header)) return 1;
    binary_reader reader(is, &header);
    bool first = true; uint64_t ppos = 0; mer_dna pkey; uint64_t n = 0;
    while(reader.next()) {
      uint64_t pos = reader.pos();
      if(!first && (pos < ppos || (pos == ppos && !(pkey < reader.key())))) {
        std::cout << "ORDER VIOLATION at record " << n << "\n"; return 2;
      }
      ppos = pos; pkey = reader.key(); first = false; ++n;
    }
    std::cout << "ORDER OK " << n << "\n";
    return 0;
  }
  char spacer = tab ? '\t' : ' ';
  std::ifstream is; jellyfish::file_header header;
  if(!open_db(db, is, header)) return 1;
  if(header.format() == binary_dumper::format)    { binary_reader r(is, &header); dump_loop(r, column, spacer, lower, upper); }
  else if(header.format() == text_dumper::format) { text_reader r(is, &header);   dump_loop(r, column, spacer, lower, upper); }
  else { std::cerr << "Unknown format '" << header.format() << "'\n"; return 1; }
  return 0;
}

template<typename Reader>
static void histo_loop(Reader& reader, uint64_t base, uint64_t ceil, uint64_t inc, std::vector<uint64_t>& histo) {
  while(reader.next()) {
    if(reader.val() < base)      ++histo[0];
    else if(reader.val() > ceil) ++histo[histo.size() - 1];
    else                         ++histo[(reader.val() - base) / inc];
  }
}

// histo [-l low] [-h high] [-i inc] [-f] db   (histo_main.cc:47-90)
static int do_histo(int argc, char* argv[]) {
  uint64_t low = 1, high = 10000, incr = 1; bool full = false; const char* db = 0;
  for(int i = 1; i < argc; ++i) {
    std::string a(argv[i]);
    if(a == "-l") low = strtoull(argv[++i], 0, 10);
    else if(a == "-h") high = strtoull(argv[++i], 0, 10);
    else if(a == "-i") incr = strtoull(argv[++i], 0, 10);
    else if(a == "-f") full = true;
    else db = argv[i];
  }
  if(!db) return 1;
  std::ifstream is; jellyfish::file_header header;
  if(!open_db(db, is, header)) return 1;
  const uint64_t base = incr >= low ? 0 : low - incr;
  const uint64_t ceil = high + incr;
  const uint64_t nb_buckets = (ceil + incr - base) / incr;
  std::vector<uint64_t> histo(nb_buckets, 0);
  if(header.format() == binary_dumper::format) { binary_reader r(is, &header); histo_loop(r, base, ceil, incr, histo); }
  else                                         { text_reader r(is, &header);   histo_loop(r, base, ceil, incr, histo); }
  uint64_t col = base;
  for(uint64_t i = 0; i < nb_buckets; ++i, col += incr)
    if(histo[i] > 0 || full) std::cout << col << " " << histo[i] << "\n";
  return 0;
}

// stats db   (stats_main.cc:33-79)
static int do_stats(int argc, char* argv[]) {
  uint64_t low = 0, high = std::numeric_limits<uint64_t>::max(); const char* db = 0;
  for(int i = 1; i < argc; ++i) {
    std::string a(argv[i]);
    if(a == "-L") low = strtoull(argv[++i], 0, 10);
    else if(a == "-U") high = strtoull(argv[++i], 0, 10);
    else db = argv[i];
  }
  if(!db) return 1;
  std::ifstream is; jellyfish::file_header header;
  if(!open_db(db, is, header)) return 1;
  uint64_t uniq = 0, distinct = 0, total = 0, max = 0;
  binary_reader reader(is, &header);
  while(reader.next()) {
    if(reader.val() < low || reader.val() > high) continue;
    uniq += reader.val() == 1; total += reader.val();
    max = std::max(max, reader.val()); ++distinct;
  }
  std::cout << "Unique:    " << uniq << "\n" << "Distinct:  " << distinct << "\n"
            << "Total:     " << total << "\n" << "Max_count: " << max << "\n";
  return 0;
}

// digest db : the content checksum of a binary/sorted or text/sorted file
static int do_digest(int argc, char* argv[]) {
  uint64_t low = 0, high = std::numeric_limits<uint64_t>::max(); const char* db = 0;
  for(int i = 1; i < argc; ++i) {
    std::string a(argv[i]);
    if(a == "-L") low = strtoull(argv[++i], 0, 10);
    else if(a == "-U") high = strtoull(argv[++i], 0, 10);
    else db = argv[i];
  }
  if(!db) return 1;
  std::ifstream is; jellyfish::file_header header;
  if(!open_db(db, is, header)) return 1;
  digest_t d;
  if(header.format() == binary_dumper::format) { binary_reader r(is, &header); while(r.next()) if(r.val() >= low && r.val() <= high) d.add(r.key(), r.val()); }
  else                                         { text_reader r(is, &header);   while(r.next()) if(r.val() >= low && r.val() <= high) d.add(r.key(), r.val()); }
  d.print(std::cout);
  return 0;
}

// query db mer...   (query_main.cc:56-70,104-115) -- random access through the
// reference's binary_query (interpolation search on the header's matrix).
static int do_query(int argc, char* argv[]) {
  if(argc < 2) return 1;
  const char* db = argv[1];
  std::ifstream in(db, std::ios::in | std::ios::binary);
  jellyfish::file_header header(in);
  if(!in.good()) { std::cerr << "Failed to parse header\n"; return 1; }
  mer_dna::k(header.key_len() / 2);
  jellyfish::mapped_file binary_map(db);
  binary_query bq(binary_map.base() + header.offset(), header.key_len(), header.counter_len(), header.matrix(),
                  header.size() - 1, binary_map.length() - header.offset());
  mer_dna m;
  auto one = [&](const std::string& s) {
    try {
      m = s;
      if(header.canonical()) m.canonicalize();
      std::cout << m << " " << bq.check(m) << "\n";
    } catch(std::length_error& e) { std::cerr << "Invalid mer '" << s << "'\n"; }
  };
  if(argc == 2) { std::string line; while(std::getline(std::cin, line)) one(line); }
  for(int i = 2; i < argc; ++i) one(argv[i]);
  return 0;
}

// header db : print the JSON header (info_main.cc --json equivalent)
static int do_header(int argc, char* argv[]) {
  if(argc < 2) return 1;
  std::ifstream is; jellyfish::file_header header;
  if(!open_db(argv[1], is, header)) return 1;
  std::cout << header.root().toStyledString();
  return 0;
}

int main(int argc, char* argv[]) {
  if(argc < 2) { std::cerr << "usage: ref_jf <count|bc|dump|histo|stats|query|header|digest> ...\n"; return 1; }
  std::string cmd(argv[1]);
  try {
    if(cmd == "count")  return do_count(argc - 1, argv + 1);
    if(cmd == "dump")   return do_dump(argc - 1, argv + 1);
    if(cmd == "histo")  return do_histo(argc - 1, argv + 1);
    if(cmd == "stats")  return do_stats(argc - 1, argv + 1);
    if(cmd == "query")  return do_query(argc - 1, argv + 1);
    if(cmd == "header") return do_header(argc - 1, argv + 1);
    if(cmd == "bc")     return do_bc(argc - 1, argv + 1);
    if(cmd == "digest") return do_digest(argc - 1, argv + 1);
  } catch(std::exception& e) {
    std::cerr << "ref_jf: " << e.what() << "\n";
    return 1;
  }
  std::cerr << "unknown command " << cmd << "\n";
  return 1;
}
