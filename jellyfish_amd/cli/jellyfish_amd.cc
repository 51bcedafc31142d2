// jellyfish_amd/cli/jellyfish_amd.cc -- `jellyfish-amd <cmd>`: the reference's CLI verbs on
// top of the MI355X engine.
//
//   count  sub_commands/count_main.cc:218-385 (+ count_main_cmdline.yaggo for the options)
//   dump   sub_commands/dump_main.cc:36-88
//   histo  sub_commands/histo_main.cc:34-90
//   stats  sub_commands/stats_main.cc:33-79
//   query  sub_commands/query_main.cc:44-123
//   info   sub_commands/info_main.cc:31-54
// Dispatch like sub_commands/jellyfish.cc:138-158.  The option parsers are written by hand
// (the reference generates them with yaggo); names, defaults and output text follow the
// reference so scripts keep working.  dump/histo/stats/query/info are plain sequential file
// readers (I/O bound, host only); count drives the GPU through the hash_counter facade.
#include <algorithm>
#include <cerrno>
#include <chrono>
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <limits>
#include <memory>
#include <thread>
#include <string>
#include <vector>

#include <jellyfish_amd/dumpers.hpp>
#include <sys/stat.h>
#include <sys/wait.h>
#include <jellyfish_amd/sequence_parser.hpp>
#include <jellyfish_amd/device_parser.hpp>

using namespace jellyfish_amd;

namespace {

[[noreturn]] void die(const std::string& msg) {   // err::die (include/jellyfish/err.hpp:77-82)
  std::cerr << msg << std::endl;
  exit(1);
}

// yaggo "suffix" numbers: k, M, G, T, ... = powers of 1000
uint64_t parse_suffix(const std::string& s, const char* opt) {
  char* end = nullptr;
  const double v = strtod(s.c_str(), &end);
  double mult = 1;
  if(end && *end) {
    switch(*end) {
    case 'k': mult = 1e3; break; case 'M': mult = 1e6; break; case 'G': mult = 1e9; break;
    case 'T': mult = 1e12; break; case 'P': mult = 1e15; break; case 'E': mult = 1e18; break;
    default: die(std::string("Invalid numeric suffix in option ") + opt + " '" + s + "'");
    }
    if(end[1]) die(std::string("Invalid number for option ") + opt + " '" + s + "'");
  }
  return (uint64_t)(v * mult);
}

struct ArgCursor {
  int argc; char** argv; int i = 1;
  bool more() const { return i < argc; }
  std::string cur() const { return argv[i]; }
  // value of an option given as "-x VAL", "-xVAL", "--long VAL" or "--long=VAL"
  std::string value(const std::string& shortf, const std::string& longf) {
    std::string a = argv[i];
    if(!longf.empty() && a.rfind(longf + "=", 0) == 0) return a.substr(longf.size() + 1);
    if(!shortf.empty() && a.size() > shortf.size() && a.rfind(shortf, 0) == 0 && a[1] != '-') return a.substr(shortf.size());
    if(i + 1 >= argc) die("Missing argument for option " + a);
    return argv[++i];
  }
  bool is(const std::string& shortf, const std::string& longf) const {
    std::string a = argv[i];
    if(!longf.empty() && (a == longf || a.rfind(longf + "=", 0) == 0)) return true;
    if(!shortf.empty() && a.rfind(shortf, 0) == 0 && (a.size() == shortf.size() || a[1] != '-')) return true;
    return false;
  }
};

// A positional argument, or an error for anything that looks like an option the verb does not know (yaggo does the same).
std::string positional(const ArgCursor& a) {
  const std::string c = a.cur();
  if(c.size() > 1 && c[0] == '-') die("Unknown option '" + c + "'");
  return c;
}

double seconds_since(std::chrono::steady_clock::time_point t0) {
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// ---------------------------------------------------------------- readers
struct db_file {
  std::ifstream is;
  file_header header;
  bool ok = true;
  explicit db_file(const std::string& path, bool die_on_error = true) : is(path, std::ios::binary) {
    if(!is.good()) { if(die_on_error) die("Failed to open input file '" + path + "'"); ok = false; return; }
    if(!header.read(is)) { if(die_on_error) die("Failed to parse header of file '" + path + "'"); ok = false; return; }
    mer_dna::k(header.key_len() / 2);
  }
};

template <typename F> void for_each_record(db_file& db, F f) {
  if(db.header.format() == binary_dumper::format) { binary_reader r(db.is, &db.header); while(r.next()) f(r.key(), r.val()); }
  else if(db.header.format() == text_dumper::format) { text_reader r(db.is, &db.header); while(r.next()) f(r.key(), r.val()); }
  else die("Unknown format '" + db.header.format() + "'");
}

// ---------------------------------------------------------------- merge  (jellyfish/merge_files.cc:36-176, sub_commands/merge_main.cc)
// k-way merge of sorted databases written with the same hash function: records of every input come in (pos, key)
// order, so a heap of the inputs' heads yields the union in that order; equal keys are combined (sum / min / max)
// or only tallied (Jaccard).  Host code: the inputs are files.
struct MergeError : public std::runtime_error { explicit MergeError(const std::string& s) : std::runtime_error(s) {} };

template <typename Reader, typename Writer>
static void do_merge(std::vector<std::unique_ptr<db_file>>& files, std::ostream& out, Writer&& write, uint64_t min, uint64_t max, int op) {
  struct head { uint64_t pos; mer_dna key; uint64_t val; size_t src; };
  const header_matrix m = files[0]->header.matrix();
  const uint64_t mask = files[0]->header.size() - 1;
  std::vector<std::unique_ptr<Reader>> readers;
  auto later = [](const head& a, const head& b) { return a.pos != b.pos ? a.pos > b.pos : a.key > b.key; };   // min-heap on (pos, key)
  std::vector<head> heap;
  auto push = [&](size_t i) {
    Reader& r = *readers[i];
    if(!r.next()) return;
    heap.push_back(head{m.times(r.key().data()) & mask, r.key(), r.val(), i});
    std::push_heap(heap.begin(), heap.end(), later);
  };
  for(size_t i = 0; i < files.size(); ++i) { readers.emplace_back(new Reader(files[i]->is, &files[i]->header)); push(i); }
  const uint64_t nb_files = files.size();
  uint64_t inter = 0, winter = 0, union_ = 0, wunion = 0;
  while(!heap.empty()) {
    const mer_dna key = heap.front().key;
    uint64_t sum = 0, maxc = 0, minc = std::numeric_limits<uint64_t>::max(), present = 0;
    while(!heap.empty() && heap.front().key == key) {
      std::pop_heap(heap.begin(), heap.end(), later);
      const head h = heap.back(); heap.pop_back();
      ++present; sum += h.val; minc = std::min(minc, h.val); maxc = std::max(maxc, h.val);
      push(h.src);
    }
    if(present < nb_files) minc = 0;               // absent from some file: count 0 there
    if(op != 3) {
      const uint64_t val = op == 0 ? sum : op == 1 ? minc : maxc;
      if(val >= min && val <= max) write(out, key, val);
    } else { inter += minc > 0; winter += minc; union_ += 1; wunion += maxc; }
  }
  if(op == 3) out << "Jaccard  " << (double)inter / (double)union_ << '\n' << "wJaccard " << (double)winter / (double)wunion << '\n';
}

// op: 0 sum, 1 min, 2 max, 3 Jaccard
static void merge_files(const std::vector<std::string>& inputs, const std::string& out_file, file_header& out_header, uint64_t min, uint64_t max, int op) {
  std::vector<std::unique_ptr<db_file>> files;
  unsigned key_len = 0, out_counter_len = std::numeric_limits<unsigned>::max();
  size_t max_reprobe_offset = 0, size = 0;
  std::string format;
  header_matrix matrix;
  for(size_t i = 0; i < inputs.size(); ++i) {
    std::unique_ptr<db_file> f(new db_file(inputs[i], false));
    if(!f->is.good() || !f->ok) throw MergeError("Failed to open input file '" + inputs[i] + "'");
    file_header& h = f->header;
    if(i == 0) {
      key_len = h.key_len(); max_reprobe_offset = h.max_reprobe_offset(); size = h.size(); matrix = h.matrix(); format = h.format();
      out_header.size(size); out_header.key_len(key_len); out_header.matrix(matrix);
      out_header.max_reprobe(h.max_reprobe()); out_header.set_reprobes(h.get_reprobes());
      out_counter_len = std::min(out_counter_len, h.counter_len());
    } else {
      if(format != h.format()) throw MergeError("Can't merge files with different formats (" + format + ", " + h.format() + ")");
      if(h.key_len() != key_len) throw MergeError("Can't merge hashes of different key lengths (" + std::to_string(key_len) + ", " + std::to_string(h.key_len()) + ")");
      if(h.max_reprobe_offset() != max_reprobe_offset) throw MergeError("Can't merge hashes with different reprobing strategies");
      if(h.size() != size) throw MergeError("Can't merge hash with different size (" + std::to_string(size) + ", " + std::to_string(h.size()) + ")");
      const header_matrix hm = h.matrix();
      if(hm.r != matrix.r || hm.c != matrix.c || hm.identity != matrix.identity || hm.columns != matrix.columns)
        throw MergeError("Can't merge hash with different hash function");
      out_counter_len = std::min(out_counter_len, h.counter_len());
    }
    files.push_back(std::move(f));
  }
  mer_dna::k(key_len / 2);
  std::ofstream out(out_file, std::ios::binary | std::ios::trunc);
  if(!out.good()) throw MergeError("Can't open out file '" + out_file + "'");
  if(op != 3) out_header.format(format);
  if(format == binary_dumper::format) {
    out_header.counter_len(out_counter_len);
    if(op != 3) out_header.write(out);
    const unsigned kb = (key_len + 7) / 8;
    const uint64_t vmax = out_counter_len >= 8 ? ~(uint64_t)0 : (((uint64_t)1 << (8 * out_counter_len)) - 1);
    do_merge<binary_reader>(files, out, [&](std::ostream& o, const mer_dna& k, uint64_t v) {
      o.write((const char*)k.data(), kb);                                  // binary_writer::write, binary_dumper.hpp:36-40
      const uint64_t w = std::min(v, vmax);
      o.write((const char*)&w, out_counter_len);
    }, min, max, op);
  } else if(format == text_dumper::format) {
    if(op != 3) out_header.write(out);
    do_merge<text_reader>(files, out, [&](std::ostream& o, const mer_dna& k, uint64_t v) { o << k << ' ' << v << '\n'; }, min, max, op);
  } else throw MergeError("Unknown format '" + format + "'");
  out.close();
}

int merge_main(int argc, char* argv[]) {
  file_header out_header;
  out_header.fill_standard();
  out_header.set_cmdline(argc, argv);
  std::string output = "mer_counts_merged.jf";
  bool min_flag = false, max_flag = false, jaccard = false, lower_given = false, upper_given = false;
  uint64_t lower = 0, upper = std::numeric_limits<uint64_t>::max();
  std::vector<std::string> inputs;
  ArgCursor a{argc, argv};
  for(; a.more(); ++a.i) {
    if(a.is("-o", "--output")) output = a.value("-o", "--output");
    else if(a.cur() == "-m" || a.cur() == "--min") min_flag = true;
    else if(a.cur() == "-M" || a.cur() == "--max") max_flag = true;
    else if(a.cur() == "-j" || a.cur() == "--jaccard") jaccard = true;
    else if(a.is("-L", "--lower-count")) { lower = strtoull(a.value("-L", "--lower-count").c_str(), 0, 10); lower_given = true; }
    else if(a.is("-U", "--upper-count")) { upper = strtoull(a.value("-U", "--upper-count").c_str(), 0, 10); upper_given = true; }
    else if(a.cur().size() > 1 && a.cur()[0] == '-') die("Unknown option '" + a.cur() + "'");
    else inputs.push_back(a.cur());
  }
  if(min_flag && max_flag) die("Switches -m, --min and -M, --max conflict");
  if(inputs.size() < 2) die("Error: at least 2 input files are required");
  const uint64_t min = lower_given ? lower : (min_flag ? 1 : 0);
  const uint64_t max = upper_given ? upper : std::numeric_limits<uint64_t>::max();
  const int op = jaccard ? 3 : max_flag ? 2 : min_flag ? 1 : 0;
  try { merge_files(inputs, output, out_header, min, max, op); } catch(MergeError& e) { die(e.what()); }
  return 0;
}

// -g / -S: every line of the file is a shell command whose standard output is a fasta / fastq stream
// (lib/generator_manager.cc runs them through named pipes; here each is read through a pipe in turn).  Blank
// lines and # comments are skipped (:223-226); a command that fails is an error.
// rank / world (count --gpus N): command j of the file is run by rank j mod N -- the commands are a partition of the input as
// good as any, every stream is read whole by one rank.
static void feed_generators(const std::string& generator, std::string shell, unsigned mer_len, const sequence_parser::sink_type& sink,
                            unsigned rank = 0, unsigned world = 1) {
  std::ifstream gf(generator);
  if(!gf.good()) die("Can't open generator file '" + generator + "'");
  if(shell.empty()) { const char* e = getenv("SHELL"); shell = e && *e ? e : "/bin/sh"; }
  sequence_parser parser(mer_len);
  std::string cmd;
  unsigned n_cmd = 0;
  while(std::getline(gf, cmd)) {
    const size_t first = cmd.find_first_not_of(" \t\n\v\f\r");
    if(first == std::string::npos || cmd[first] == '#') continue;
    if(n_cmd++ % world != rank) continue;
    int fds[2];
    if(pipe(fds) != 0) die("pipe() failed");
    const pid_t pid = fork();
    if(pid < 0) die("fork() failed");
    if(pid == 0) {
      close(fds[0]); dup2(fds[1], 1); close(fds[1]);
      execl(shell.c_str(), shell.c_str(), "-c", cmd.c_str(), (char*)0);
      _exit(127);
    }
    close(fds[1]);
    try { parser.parse_stream(fds[0], sink); }          // as it arrives, in pieces of whole records
    catch(std::exception& e) { close(fds[0]); waitpid(pid, nullptr, 0); throw; }
    close(fds[0]);
    int status = 0;
    waitpid(pid, &status, 0);
    if(!WIFEXITED(status) || WEXITSTATUS(status) != 0) die("Generator command failed: " + cmd);
  }
}

// ---------------------------------------------------------------- count
// ---- count --gpus N: one process per GPU -----------------------------------------------------------
// The command line that carries --gpus N starts N copies of itself (environment: JFGPU_RANK, JFGPU_WORLD,
// JFGPU_RENDEZVOUS = a scratch directory) and waits for them; each copy opens GPU `rank`, holds shard `rank` of the table,
// reads its part of every input file, routes its k-mers to the owning shards over RCCL (jfgpu_comm_*) and writes its records
// at its own offset of the common output file.  A launcher that sets RANK / WORLD_SIZE / LOCAL_RANK itself (torchrun, mpirun
// wrappers) and JFGPU_RENDEZVOUS is taken at its word: no copies are started.
struct rank_env { int rank = 0, world = 1, local_rank = 0; std::string rendezvous; bool is_rank = false; };

rank_env read_rank_env() {
  rank_env e;
  const char* rdv = getenv("JFGPU_RENDEZVOUS");
  const char* r = getenv("JFGPU_RANK");
  const char* w = getenv("JFGPU_WORLD");
  // a launcher's generic RANK / WORLD_SIZE count only together with JFGPU_RENDEZVOUS: `count --gpus 2` typed inside some
  // torchrun / SLURM shell must start its own ranks, not become one rank waiting for peers that never come
  if(rdv && (!r || !w)) { r = getenv("RANK"); w = getenv("WORLD_SIZE"); }
  if(!r || !w) return e;
  e.is_rank = true; e.rank = atoi(r); e.world = atoi(w);
  const char* l = getenv("JFGPU_LOCAL_RANK"); if(!l) l = getenv("LOCAL_RANK");
  e.local_rank = l ? atoi(l) : e.rank;
  if(rdv) e.rendezvous = rdv;
  return e;
}

int spawn_ranks(unsigned n, char* argv[]) {
  char tmpl[] = "/tmp/jfgpu_rdv_XXXXXX";
  const char* dir = mkdtemp(tmpl);
  if(!dir) die("Can't create the rendezvous directory under /tmp");
  std::vector<pid_t> pids;
  for(unsigned r = 0; r < n; ++r) {
    const pid_t pid = fork();
    if(pid < 0) die("fork failed");
    if(pid == 0) {
      setenv("JFGPU_RANK", std::to_string(r).c_str(), 1);
      setenv("JFGPU_WORLD", std::to_string(n).c_str(), 1);
      setenv("JFGPU_LOCAL_RANK", std::to_string(r).c_str(), 1);
      setenv("JFGPU_RENDEZVOUS", dir, 1);
      std::vector<char*> av;                       // argv here starts at the verb ("count", ...): put the program name back
      av.push_back(const_cast<char*>("jellyfish-amd"));
      for(char** a = argv; *a; ++a) av.push_back(*a);
      av.push_back(nullptr);
      execv("/proc/self/exe", av.data());
      perror("execv");
      _exit(127);
    }
    pids.push_back(pid);
  }
  // The ranks meet in collectives: when one of them dies ("Hash full", an unreadable file, no memory ...) the others would
  // wait for it forever, so the first failure ends them all and is what this command returns.
  int worst = 0;
  size_t left = pids.size();
  while(left) {
    int st = 0;
    const pid_t pid = waitpid(-1, &st, 0);
    if(pid < 0) { if(errno == EINTR) continue; worst = std::max(worst, 1); break; }
    if(std::find(pids.begin(), pids.end(), pid) == pids.end()) continue;
    --left;
    const int rc = WIFEXITED(st) ? WEXITSTATUS(st) : 128 + (WIFSIGNALED(st) ? WTERMSIG(st) : 0);
    if(rc != 0 && worst == 0) {
      worst = rc;
      for(pid_t q : pids) if(q != pid) kill(q, SIGTERM);
    }
  }
  unlink((std::string(dir) + "/id").c_str());
  rmdir(dir);
  return worst;
}

// rank 0 makes the RCCL id and leaves it in the rendezvous directory; the others wait for it
void exchange_unique_id(const rank_env& e, uint8_t* id128) {
  const std::string path = e.rendezvous + "/id", tmp = path + ".tmp";
  if(e.rank == 0) {
    if(jfgpu_comm_unique_id(id128)) die(jfgpu_last_error());
    unlink(path.c_str());                                  // (a file left by an earlier job in a launcher-provided directory)
    std::ofstream out(tmp, std::ios::binary | std::ios::trunc);
    out.write((const char*)id128, 128);
    out.close();
    if(!out.good() || rename(tmp.c_str(), path.c_str()) != 0) die("Can't write '" + path + "'");
    return;
  }
  for(int tries = 0; tries < 6000; ++tries) {            // 10 minutes: rank 0 may still be paging its libraries in
    std::ifstream in(path, std::ios::binary);
    if(in.good()) { in.read((char*)id128, 128); if(in.gcount() == 128) return; }
    std::this_thread::sleep_for(std::chrono::milliseconds(100));
  }
  die("Timed out waiting for rank 0 in '" + e.rendezvous + "'");
}

int count_main(int argc, char* argv[]) {
  auto start_time = std::chrono::steady_clock::now();
  file_header header;
  header.fill_standard();
  header.set_cmdline(argc, argv);

  unsigned mer_len = 0, threads = 1, counter_len = 7, out_counter_len = 4, reprobes = 126, Files = 1;
  uint64_t size = 0, lower = 0, upper = std::numeric_limits<uint64_t>::max();
  bool counter_len_given = false, reprobes_given = false, generators_given = false;
  bool size_given = false, lower_given = false, upper_given = false, canonical = false, text = false, no_write = false, disk = false, host_parse = false, no_merge = false, no_unlink = false;
  uint64_t bf_size = 0; double bf_fp = 0.01; bool bf_size_given = false;
  int device = -1, min_qual = 0, quality_start = 64, min_quality = 0;
  unsigned gpus = 1; bool gpus_given = false;
  bool min_qual_char_given = false, min_quality_given = false;
  uint32_t matrix_kind = JFGPU_MATRIX_DEFAULT;
  std::string output = "mer_counts.jf", timing, bc_path, generator, shell, digest_path;
  std::vector<std::string> files, if_files;
  ArgCursor a{argc, argv};
  for(; a.more(); ++a.i) {
    if(a.is("-m", "--mer-len")) mer_len = (unsigned)strtoul(a.value("-m", "--mer-len").c_str(), 0, 10);
    else if(a.is("-s", "--size")) { size = parse_suffix(a.value("-s", "--size"), "-s"); size_given = true; }
    else if(a.is("-t", "--threads")) threads = (unsigned)strtoul(a.value("-t", "--threads").c_str(), 0, 10);
    else if(a.is("", "--if")) if_files.push_back(a.value("", "--if"));
    else if(a.is("-g", "--generator")) generator = a.value("-g", "--generator");
    else if(a.is("-G", "--Generators")) { (void)a.value("-G", "--Generators"); generators_given = true; }   // generators run one after the other here
    else if(a.is("-S", "--shell")) shell = a.value("-S", "--shell");
    else if(a.is("-F", "--Files")) Files = (unsigned)strtoul(a.value("-F", "--Files").c_str(), 0, 10);
    else if(a.is("-c", "--counter-len")) { counter_len = (unsigned)strtoul(a.value("-c", "--counter-len").c_str(), 0, 10); counter_len_given = true; }
    else if(a.is("", "--out-counter-len")) out_counter_len = (unsigned)strtoul(a.value("", "--out-counter-len").c_str(), 0, 10);
    else if(a.is("-p", "--reprobes")) { reprobes = (unsigned)strtoul(a.value("-p", "--reprobes").c_str(), 0, 10); reprobes_given = true; }
    else if(a.is("-o", "--output")) output = a.value("-o", "--output");
    else if(a.is("-L", "--lower-count")) { lower = strtoull(a.value("-L", "--lower-count").c_str(), 0, 10); lower_given = true; }
    else if(a.is("-U", "--upper-count")) { upper = strtoull(a.value("-U", "--upper-count").c_str(), 0, 10); upper_given = true; }
    else if(a.is("", "--timing")) timing = a.value("", "--timing");
    else if(a.is("", "--device")) device = atoi(a.value("", "--device").c_str());
    else if(a.is("", "--gpus")) { gpus = (unsigned)strtoul(a.value("", "--gpus").c_str(), 0, 10); gpus_given = true; }
    else if(a.is("", "--bc")) bc_path = a.value("", "--bc");
    else if(a.is("", "--matrix")) {                          // hash matrix family (jfgpu.h: JFGPU_MATRIX_*); the file header carries the matrix either way
      const std::string v = a.value("", "--matrix");
      if(v == "xs" || v == "xorshift") matrix_kind = JFGPU_MATRIX_XORSHIFT;
      else if(v == "reference") matrix_kind = JFGPU_MATRIX_REFERENCE;
      else die("--matrix must be xs or reference");
    }
    else if(a.is("", "--digest")) digest_path = a.value("", "--digest");   // content checksum of the table (jfgpu_digest), for at-scale parity checks
    else if(a.cur() == "-C" || a.cur() == "--canonical") canonical = true;
    else if(a.cur() == "--text") text = true;
    else if(a.cur() == "--no-write") no_write = true;
    else if(a.cur() == "--host-parse") host_parse = true;   // read the files with the host reader instead of the device parser
    else if(a.cur() == "--disk") disk = true;   // do_size_doubling(false) (count_main.cc:276-277): a full table is written out as a sorted run
    else if(a.cur() == "--no-merge") no_merge = true;
    else if(a.cur() == "--no-unlink") no_unlink = true;
    else if(a.is("-Q", "--min-qual-char")) {             // count_main.cc:234-244
      const std::string v = a.value("-Q", "--min-qual-char");
      if(v.size() != 1) die("[-Q, --min-qual-char] must be one character.");
      if(v[0] < '!' || v[0] > '~') die(std::string("Quality character '") + v + "' is outside of the range [!, ~]");
      min_qual = v[0]; min_qual_char_given = true;
    }
    else if(a.is("", "--quality-start")) quality_start = atoi(a.value("", "--quality-start").c_str());
    else if(a.is("", "--min-quality")) { min_quality = atoi(a.value("", "--min-quality").c_str()); min_quality_given = true; }
    else if(a.is("", "--bf-size")) { bf_size = parse_suffix(a.value("", "--bf-size"), "--bf-size"); bf_size_given = true; }
    else if(a.is("", "--bf-fp")) bf_fp = atof(a.value("", "--bf-fp").c_str());
    else if(a.is("", "--sam"))
      die("SAM/BAM/CRAM not supported (missing htslib).");
    else if(a.cur() == "-h" || a.cur() == "--help") {
      std::cout << "Usage: jellyfish-amd count [options] file:path+\n\n"
                   "Count k-mers in fasta or fastq files on an MI355X\n\n"
                   " -m, --mer-len=uint32       *Length of mer\n"
                   " -s, --size=uint64          *Initial hash size\n"
                   " -t, --threads=uint32        Number of threads (1)\n"
                   " -o, --output=string         Output file (mer_counts.jf)\n"
                   " -c, --counter-len=Length    Length bits of counting field (7)\n"
                   "     --out-counter-len=bytes Length in bytes of counter field in output (4)\n"
                   " -C, --canonical             Count both strand, canonical representation (false)\n"
                   " -p, --reprobes=uint32       Maximum number of reprobes (126)\n"
                   " -L, --lower-count=uint64    Don't output k-mer with count < lower-count\n"
                   " -U, --upper-count=uint64    Don't output k-mer with count > upper-count\n"
                   "     --if=path               Count only the k-mers of these fasta / fastq files (repeatable)\n"
                   " -g, --generator=path        File of commands generating fast[aq] (one per line, e.g. zcat reads.fa.gz)\n"
                   " -S, --shell=string          Shell used to run generator commands ($SHELL or /bin/sh)\n"
                   "     --text                  Dump in text format (false)\n"
                   "     --timing=Timing file    Print timing information\n"
                   "     --device=int            HIP device ordinal (current)\n"
                   "     --gpus=N                Spread the table over N GPUs (a power of two), one process each\n"
                   "     --host-parse            Parse the sequence files on the host (default: on the device)\n"
                   "     --matrix=xs|reference   Hash matrix family: xs = evaluated in registers by the GPU kernels (fastest);\n"
                   "                             reference = the matrix jellyfish itself draws (byte-identical files; default)\n";
      return 0;
    } else if(a.cur().size() > 1 && a.cur()[0] == '-' && a.cur() != "-") die("Unknown option '" + a.cur() + "'");
    else files.push_back(a.cur());
  }
  if(!mer_len) die("Error: mandatory switch missing: -m, --mer-len");
  if(!size_given) die("Error: mandatory switch missing: -s, --size");
  if(files.empty() && generator.empty()) die("Error: at least 1 file argument is required");
  if(bf_size_given && !bc_path.empty()) die("Switches --bf-size and --bc conflict");
  if(min_quality_given) {                                   // count_main.cc:245-256
    if(min_qual_char_given) die("Switches --min-quality and -Q, --min-qual-char conflict");
    if(quality_start < '!' || quality_start > '~') die("Quality start " + std::to_string(quality_start) + " is outside the range [33, 126]");
    min_qual = quality_start + min_quality;
    if(min_qual < '!' || min_qual > '~') die("Min quality " + std::to_string(min_quality) + " is outside the range [0, " + std::to_string((int)'~' - quality_start) + "]");
  }
  (void)threads; (void)counter_len; (void)reprobes; (void)Files;
  // Accepted for script compatibility but meaningless on this engine: say so once instead of silently ignoring them.
  // (-t: the device has its own parallelism; -c / -p: the in-memory slot format and probing are the engine's own and never
  // reach the file -- the header's val_len / max_reprobe / reprobes[] are the values readers and `merge` need to agree on.)
  if(!getenv("JFGPU_QUIET")) {
    if(counter_len_given) std::cerr << "jellyfish-amd: note: -c/--counter-len has no effect (in-memory counts are exact 64-bit)\n";
    if(reprobes_given) std::cerr << "jellyfish-amd: note: -p/--reprobes has no effect (tile-local probing on the device)\n";
    if(generators_given) std::cerr << "jellyfish-amd: note: -G/--Generators has no effect (generator commands run one after the other)\n";
  }
  if(mer_len > 128) die("jellyfish-amd: mer length > 128 (more than four key words) is not built");
  if(text) out_counter_len = 8;   // text counts are not saturated (text_dumper.hpp:18-20)

  // --gpus N: this process is either the one the user typed (it starts the ranks and waits) or one of the ranks
  rank_env renv;
  jfgpu_comm* comm = nullptr;
  uint32_t shard_bits = 0;
  if(gpus_given) {
    if(gpus < 1 || (gpus & (gpus - 1)) || gpus > 256) die("--gpus must be a power of two");
    if(mer_len > 64) die("--gpus: sharded tables for mer length > 64 are not built yet");
    if(bf_size_given || disk)
      die("--gpus cannot be combined with --bf-size (a one-pass filter cannot be sharded by input) or --disk yet");
    // (--if and --bc over shards: keys of one and two words -- every rank loads the whole counter and asks it before routing;
    //  mer length > 64 has no shards at all: refused above)
    renv = read_rank_env();
    if(!renv.is_rank) return spawn_ranks(gpus, argv);
    if(renv.world != (int)gpus || renv.rank < 0 || renv.rank >= renv.world) die("--gpus does not match the ranks' environment (WORLD_SIZE / RANK)");
    if(renv.rendezvous.empty()) die("--gpus under an external launcher: set JFGPU_RENDEZVOUS to a directory all ranks see");
    if(device < 0) device = renv.local_rank;
    { const char* tr = getenv("JFGPU_COMM_TRANSPORT");     // the test transport: every rank on the devices there are (one GPU: all on it)
      if(tr && !strcmp(tr, "ipc")) { const int nd = jfgpu_device_count(); if(nd > 0) device %= nd; } }
    while((1u << shard_bits) < gpus) ++shard_bits;
  }

  mer_dna::k(mer_len);
  header.canonical(canonical);
  // JFGPU_INIT_TRACE=1: where the Init phase goes (stderr)
  const bool init_trace = getenv("JFGPU_INIT_TRACE") != nullptr;
  auto init_mark = [&](const char* what) { if(init_trace) std::cerr << "[init] " << seconds_since(start_time) << " s  " << what << "\n"; };
  init_mark("options parsed");
  std::unique_ptr<mer_hash> ary;
  try {
    ary.reset(new mer_hash(size, mer_len * 2, counter_len, threads, reprobes, canonical, device, out_counter_len, 0, shard_bits, (uint32_t)renv.rank, matrix_kind));
  } catch(std::exception& e) { die(std::string("Failed to allocate the hash: ") + e.what()); }
  init_mark("table created (allocated and cleared)");
  if(disk) ary->do_size_doubling(false);
  if(gpus_given) {
    uint8_t id[128];
    exchange_unique_id(renv, id);
    if(jfgpu_comm_create(renv.world, renv.rank, id, device, &comm)) die(std::string("Failed to create the communicator: ") + jfgpu_last_error());
    ary->attach_comm(comm);
    if(const char* fr = getenv("JFGPU_TEST_FAIL_RANK")) if(atoi(fr) == renv.rank) die("rank told to fail (JFGPU_TEST_FAIL_RANK)");      // (tests: the siblings must not hang)
  }

  // Bloom counter read from file to filter out low frequency k-mers, two pass algorithm
  // (load_bloom_filter, count_main.cc:191-206,313-316)
  jfgpu_bloom* bc = nullptr;
  if(!bc_path.empty()) {
    std::ifstream in(bc_path, std::ios::in | std::ios::binary);
    file_header bh(in);
    if(!in.good()) die("Failed to parse bloom filter file '" + bc_path + "'");
    if(bh.format() != "bloomcounter") die("Invalid format '" + bh.format() + "'. Expected 'bloomcounter'");
    if(bh.key_len() != mer_len * 2) die("Invalid mer length in bloom filter");
    header_matrix m1 = bh.matrix(1), m2 = bh.matrix(2);
    jfgpu_bloom_params bp;
    memset(&bp, 0, sizeof bp);
    bp.k = mer_len; bp.canonical = canonical; bp.m = bh.size(); bp.nb_hashes = (uint32_t)bh.nb_hashes(); bp.device = device;
    bp.matrix1 = m1.columns.data(); bp.matrix2 = m2.columns.data();
    if(jfgpu_bc_create(&bp, &bc)) die(std::string("Failed to create the bloom filter: ") + jfgpu_last_error());
    std::vector<uint8_t> body(bh.size() / 5 + (bh.size() % 5 != 0));
    in.read((char*)body.data(), body.size());
    if(!in.good()) die("Bloom filter file is truncated");
    if(jfgpu_bc_load(bc, body.data()) || jfgpu_attach_bloom(ary->handle(), bc)) die(jfgpu_last_error());
  }

  // Bloom filter to filter out low frequency k-mers, one pass algorithm (count_main.cc:318-324): the first sighting of
  // a k-mer only marks it in the filter, later ones are counted
  if(bf_size_given) {
    jfgpu_bloom_params bp;
    memset(&bp, 0, sizeof bp);
    bp.k = mer_len; bp.canonical = canonical; bp.device = device;
    bp.m = jfgpu_bc_opt_m(bf_fp, bf_size); bp.nb_hashes = jfgpu_bc_opt_k(bf_fp);
    if(jfgpu_bf_create(&bp, &bc)) die(std::string("Failed to create the bloom filter: ") + jfgpu_last_error());
    if(jfgpu_attach_bloom(ary->handle(), bc)) die(jfgpu_last_error());
  }

  std::unique_ptr<dumper_base> dumper;
  if(text) dumper.reset(new text_dumper(threads, output.c_str(), &header));
  else dumper.reset(new binary_dumper(out_counter_len, ary->key_len(), threads, output.c_str(), &header));
  if(disk && !no_write) {      // intermediate sorted runs <output>0, <output>1, ... (dumper.hpp:45-61), merged at the end
    dumper->one_file(false);
    ary->on_full([&]() { dumper->dump(ary->ary()); });
  }
  // Still "Init" (count_main.cc:286: the reference allocates and touches its table before the clock of the Counting phase
  // starts): the device workspace for the whole input (file sizes are an upper bound of the sequence) and the feed's
  // pinned staging buffers.  Tens of GB of fresh device memory cost seconds to hand out, whoever asks first.
  std::unique_ptr<device_sequence_parser> dev_parser;
  {
    uint64_t total = 0, largest = 0;
    for(const auto& f : files) { struct stat st; if(stat(f.c_str(), &st) == 0 && S_ISREG(st.st_mode)) { total += (uint64_t)st.st_size; largest = std::max<uint64_t>(largest, st.st_size); } }
    if(gpus_given) { total = total / gpus + ((uint64_t)1 << 20); largest = largest / gpus + ((uint64_t)1 << 20); }   // this rank's part
    if(!host_parse && total > ((uint64_t)64 << 20)) ary->expect_input(std::min<uint64_t>(total, (uint64_t)12 << 30));
    init_mark("partition workspace reserved");
    if(!host_parse) {
      try { dev_parser.reset(new device_sequence_parser(mer_len, device)); dev_parser->min_quality(min_qual); dev_parser->prepare(largest); }
      catch(std::exception& e) { die(e.what()); }
    }
  }
  init_mark("feed prepared (pinned buffers, device staging)");
  const double init_s = seconds_since(start_time);

  auto count_start = std::chrono::steady_clock::now();
  double parse_ms = 0; size_t fallback_bytes = 0;
  auto feed = [&](const std::vector<std::string>& paths) {
    if(host_parse) {
      sequence_parser parser(mer_len);
      parser.min_quality(min_qual);
      // (--gpus N: file i is read by rank i mod N -- whole files are a partition of the input too; a single file is one rank's)
      for(size_t i = 0; i < paths.size(); ++i)
        if(!gpus_given || i % gpus == (size_t)renv.rank)
          parser.parse_file(paths[i].c_str(), [&](const char* buf, size_t n) { ary->count_sequence(buf, n); });
    } else {
      device_sequence_parser& parser = *dev_parser;
      const double ms0 = parser.device_ms(); const size_t fb0 = parser.host_fallback_bytes();
      // (JFGPU_TEST_PARTS=N: read every file as N parts, one after the other -- what N ranks would read between them)
      const unsigned test_parts = !gpus_given && getenv("JFGPU_TEST_PARTS") ? std::max(1, atoi(getenv("JFGPU_TEST_PARTS"))) : 1u;
      for(const auto& f : paths)
        for(unsigned part = 0; part < test_parts; ++part)
          parser.parse_file_part(f.c_str(), gpus_given ? (unsigned)renv.rank : part, gpus_given ? gpus : test_parts,
                                 [&](const char* d_buf, size_t n) { ary->count_sequence_dev(d_buf, n); },
                                 [&](const char* buf, size_t n) { ary->count_sequence(buf, n); }, [&]() { ary->wait_consumed(); });
      parse_ms += parser.device_ms() - ms0; fallback_bytes += parser.host_fallback_bytes() - fb0;
    }
    ary->done();
  };
  try {
    if(!if_files.empty()) {   // count_main.cc:289-295: prime the hash with the --if mers, then only update
      ary->set_operation(mer_hash::PRIME);
      feed(if_files);
      ary->set_operation(mer_hash::UPDATE);
    }
    feed(files);
    if(!generator.empty()) {
      feed_generators(generator, shell, mer_len, [&](const char* buf, size_t n) { ary->count_sequence(buf, n); },
                      gpus_given ? (unsigned)renv.rank : 0u, gpus_given ? gpus : 1u);
      ary->done();
    }
  } catch(std::exception& e) { die(e.what()); }
  const double count_s = seconds_since(count_start);

  if(!digest_path.empty()) {
    uint64_t d[4];
    if(jfgpu_digest(ary->handle(), lower_given ? lower : 0, upper_given ? upper : std::numeric_limits<uint64_t>::max(), d)) die(jfgpu_last_error());
    if(comm) {                                     // the digest of the whole table: records, totals and sums add up, the xor words xor
      std::vector<uint64_t> all(renv.world);
      for(int i = 0; i < 4; ++i) {
        if(jfgpu_comm_allgather_u64(comm, d[i], all.data())) die(jfgpu_last_error());
        uint64_t v = 0;
        for(uint64_t x : all) v = i == 3 ? (v ^ x) : (v + x);
        d[i] = v;
      }
    }
    if(renv.rank == 0) {
      std::ofstream df(digest_path);
      df << "records " << d[0] << "\ntotal " << d[1] << "\nsum " << d[2] << "\nxor " << d[3] << "\n";
    }
  }
  auto write_start = std::chrono::steady_clock::now();
  if(!no_write) {
    try {
      if(dumper->nb_files() == 0) {      // no intermediate files: dump directly into the output file (count_main.cc:348-355)
        dumper->one_file(true);
        if(lower_given) dumper->min(lower);
        if(upper_given) dumper->max(upper);
        dumper->dump(ary->ary());
      } else {                           // a last run, then one round of merging (:356-371)
        dumper->dump(ary->ary());
        if(!no_merge) {
          const std::vector<std::string> parts = dumper->file_names();
          try { merge_files(parts, output, header, lower_given ? lower : 0, upper_given ? upper : std::numeric_limits<uint64_t>::max(), 0); }
          catch(MergeError& e) { die(e.what()); }
          if(!no_unlink) for(const auto& f : parts) unlink(f.c_str());
        }
      }
    } catch(std::exception& e) { die(e.what()); }
  }
  const double write_s = seconds_since(write_start);
  uint64_t ctrs[JFGPU_N_COUNTERS] = {0};
  if(!timing.empty() && getenv("JFGPU_TIMING_DETAIL")) (void)jfgpu_get_counters(ary->handle(), ctrs, JFGPU_N_COUNTERS);

  if(bc) { jfgpu_attach_bloom(ary->handle(), nullptr); jfgpu_bc_destroy(bc); }
  if(comm) {
    uint64_t done = 1;                             // every rank's records are in the file before anyone reports success
    if(jfgpu_comm_allreduce_u64(comm, &done, 1, 0)) die(jfgpu_last_error());
    ary->attach_comm(nullptr);
    jfgpu_comm_destroy(comm);
  }

  if(!timing.empty() && renv.rank == 0) {   // count_main.cc:375-382
    std::ofstream tf(timing);
    tf << "Init     " << init_s << "\n"
       << "Counting " << count_s << "\n"
       << "Writing  " << write_s << "\n";
    if(!host_parse && getenv("JFGPU_TIMING_DETAIL"))     // extra lines only on request: the file keeps the reference's three
      tf << "DeviceParse " << parse_ms / 1e3 << "\n" << "HostParsedBytes " << fallback_bytes << "\n"
         << "DirectInserts " << ctrs[5] << "\n" << "FlushesPlain " << ctrs[8] << "\n" << "FlushesHeavy " << ctrs[9] << "\n"
         << "P2Roles " << ctrs[10] << "\n" << "P2Ring " << ctrs[11] << "\n" << "P2Sort " << ctrs[12] << "\n" << "P2Exact " << ctrs[13] << "\n"
         << "P1Ring " << ctrs[14] << "\n" << "P1Other " << ctrs[15] << "\n";
  }
  return 0;
}


// ---------------------------------------------------------------- bc  (sub_commands/bc_main.cc:84-161)
int bc_main(int argc, char* argv[]) {
  auto start_time = std::chrono::steady_clock::now();
  file_header header;
  header.fill_standard();
  header.set_cmdline(argc, argv);
  unsigned mer_len = 0; uint64_t size = 0; double fpr = 0.001; bool canonical = false, size_given = false, host_parse = false;
  int device = -1;
  unsigned gpus = 1; bool gpus_given = false;
  std::string output = "mer_bloom_filter", timing, generator, shell;
  std::vector<std::string> files;
  ArgCursor a{argc, argv};
  for(; a.more(); ++a.i) {
    if(a.is("-m", "--mer-len")) mer_len = (unsigned)strtoul(a.value("-m", "--mer-len").c_str(), 0, 10);
    else if(a.is("-s", "--size")) { size = parse_suffix(a.value("-s", "--size"), "-s"); size_given = true; }
    else if(a.is("-f", "--fpr")) fpr = atof(a.value("-f", "--fpr").c_str());
    else if(a.is("-t", "--threads")) (void)a.value("-t", "--threads");
    else if(a.is("-F", "--Files")) (void)a.value("-F", "--Files");
    else if(a.is("-o", "--output")) output = a.value("-o", "--output");
    else if(a.is("", "--timing")) timing = a.value("", "--timing");
    else if(a.is("", "--device")) device = atoi(a.value("", "--device").c_str());
    else if(a.is("", "--gpus")) { gpus = (unsigned)strtoul(a.value("", "--gpus").c_str(), 0, 10); gpus_given = true; }
    else if(a.cur() == "-C" || a.cur() == "--canonical") canonical = true;
    else if(a.cur() == "--host-parse") host_parse = true;
    else if(a.is("-g", "--generator")) generator = a.value("-g", "--generator");
    else if(a.is("-G", "--Generators")) (void)a.value("-G", "--Generators");
    else if(a.is("-S", "--shell")) shell = a.value("-S", "--shell");
    else if(a.cur().size() > 1 && a.cur()[0] == '-') die("Unknown option '" + a.cur() + "'");
    else files.push_back(a.cur());
  }
  if(!mer_len) die("Error: mandatory switch missing: -m, --mer-len");
  if(!size_given) die("Error: mandatory switch missing: -s, --size");
  if(files.empty() && generator.empty()) die("Error: at least 1 file argument is required");
  if(mer_len > 64) die("jellyfish-amd: Bloom counters for mer length > 64 are not built");
  // --gpus N (like count --gpus N: this process starts the ranks, or is one): every rank inserts its part of every file into
  // its own counter, the counters are merged on the devices (jfgpu_comm_bc_merge) and rank 0 writes the file
  rank_env renv;
  if(gpus_given) {
    if(gpus < 1 || (gpus & (gpus - 1)) || gpus > 256) die("--gpus must be a power of two");
    if(host_parse || !generator.empty()) die("--gpus cannot be combined with --host-parse or -g yet");
    renv = read_rank_env();
    if(!renv.is_rank) return spawn_ranks(gpus, argv);
    if(renv.world != (int)gpus || renv.rank < 0 || renv.rank >= renv.world) die("--gpus does not match the ranks' environment (WORLD_SIZE / RANK)");
    if(renv.rendezvous.empty()) die("--gpus under an external launcher: set JFGPU_RENDEZVOUS to a directory all ranks see");
    if(device < 0) device = renv.local_rank;
    { const char* tr = getenv("JFGPU_COMM_TRANSPORT");
      if(tr && !strcmp(tr, "ipc")) { const int nd = jfgpu_device_count(); if(nd > 0) device %= nd; } }
  }
  const bool writer = renv.rank == 0;
  mer_dna::k(mer_len);
  header.canonical(canonical);
  std::ofstream out;
  if(writer) { out.open(output, std::ios::binary | std::ios::trunc); if(!out.good()) die("Can't open output file '" + output + "'"); }
  jfgpu_bloom_params bp;
  memset(&bp, 0, sizeof bp);
  bp.k = mer_len; bp.canonical = canonical; bp.device = device;
  bp.m = jfgpu_bc_opt_m(fpr, size); bp.nb_hashes = jfgpu_bc_opt_k(fpr);
  jfgpu_bloom* bc = nullptr;
  if(jfgpu_bc_create(&bp, &bc)) die(std::string("Failed to create the bloom filter: ") + jfgpu_last_error());
  uint64_t m = 0, nbytes = 0; uint32_t nh = 0;
  header_matrix m1, m2;
  m1.r = m2.r = 64; m1.c = m2.c = 2 * mer_len; m1.columns.assign(m1.c, 0); m2.columns.assign(m2.c, 0);
  jfgpu_bc_get_info(bc, &m, &nh, &nbytes, m1.columns.data(), m2.columns.data());
  header.format("bloomcounter");
  header.key_len(mer_len * 2);
  header.matrix(m1, 1);
  header.matrix(m2, 2);
  header.size(m);
  header.nb_hashes(nh);
  if(writer) header.write(out);
  jfgpu_comm* comm = nullptr;
  if(gpus_given) {
    uint8_t id[128];
    exchange_unique_id(renv, id);
    if(jfgpu_comm_create(renv.world, renv.rank, id, device, &comm)) die(std::string("Failed to create the communicator: ") + jfgpu_last_error());
    if(const char* fr = getenv("JFGPU_TEST_FAIL_RANK")) if(atoi(fr) == renv.rank) die("rank told to fail (JFGPU_TEST_FAIL_RANK)");
  }
  const double init_s = seconds_since(start_time);
  auto count_start = std::chrono::steady_clock::now();
  try {
    auto host_sink = [&](const char* buf, size_t n) { if(jfgpu_bc_insert_ascii(bc, buf, n)) throw std::runtime_error(jfgpu_last_error()); };
    if(host_parse) {
      sequence_parser parser(mer_len);
      for(const auto& f : files) parser.parse_file(f.c_str(), host_sink);
    } else {
      device_sequence_parser parser(mer_len, device);
      for(const auto& f : files)
        parser.parse_file_part(f.c_str(), gpus_given ? (unsigned)renv.rank : 0u, gpus_given ? gpus : 1u,
                               [&](const char* d_buf, size_t n) { if(jfgpu_bc_insert_ascii_dev(bc, d_buf, n)) throw std::runtime_error(jfgpu_last_error()); },
                               host_sink, [&]() { if(jfgpu_bc_sync(bc, nullptr)) throw std::runtime_error(jfgpu_last_error()); });
    }
    if(!generator.empty()) feed_generators(generator, shell, mer_len, host_sink);
    if(jfgpu_bc_sync(bc, nullptr)) throw std::runtime_error(jfgpu_last_error());
    if(comm && jfgpu_comm_bc_merge(comm, bc)) throw std::runtime_error(jfgpu_last_error());      // every rank's counter: the whole input's
  } catch(std::exception& e) { die(e.what()); }
  const double count_s = seconds_since(count_start);
  auto write_start = std::chrono::steady_clock::now();
  if(writer) {
    std::vector<uint8_t> body(nbytes);
    if(jfgpu_bc_read(bc, body.data())) die(jfgpu_last_error());
    out.write((const char*)body.data(), body.size());
    out.close();
    if(!out.good()) die("Error writing '" + output + "'");
  }
  if(comm) {
    uint64_t done = 1;                             // the file is written before anyone reports success
    if(jfgpu_comm_allreduce_u64(comm, &done, 1, 0)) die(jfgpu_last_error());
    jfgpu_comm_destroy(comm);
  }
  jfgpu_bc_destroy(bc);
  if(!timing.empty() && writer) {
    std::ofstream tf(timing);
    tf << "Init     " << init_s << "\n" << "Counting " << count_s << "\n" << "Writing  " << seconds_since(write_start) << "\n";
  }
  return 0;
}

int dump_main(int argc, char* argv[]) {
  bool column = false, tab = false;
  uint64_t lower = 0, upper = std::numeric_limits<uint64_t>::max();
  std::string output, db;
  ArgCursor a{argc, argv};
  for(; a.more(); ++a.i) {
    if(a.cur() == "-c" || a.cur() == "--column") column = true;
    else if(a.cur() == "-t" || a.cur() == "--tab") tab = true;
    else if(a.cur() == "-ct" || a.cur() == "-tc") column = tab = true;
    else if(a.is("-L", "--lower-count")) lower = strtoull(a.value("-L", "--lower-count").c_str(), 0, 10);
    else if(a.is("-U", "--upper-count")) upper = strtoull(a.value("-U", "--upper-count").c_str(), 0, 10);
    else if(a.is("-o", "--output")) output = a.value("-o", "--output");
    else db = positional(a);
  }
  if(db.empty()) die("Usage: jellyfish-amd dump [-c] [-t] [-L l] [-U u] [-o out] db:path");
  std::ios::sync_with_stdio(false);
  std::ofstream fout;
  if(!output.empty()) { fout.open(output); if(!fout.good()) die("Error opening output file '" + output + "'"); }
  std::ostream& out = output.empty() ? std::cout : fout;
  db_file f(db);
  const char spacer = tab ? '\t' : ' ';
  for_each_record(f, [&](const mer_dna& k, uint64_t v) {
    if(v < lower || v > upper) return;
    if(column) out << k << spacer << v << "\n";
    else out << ">" << v << "\n" << k << "\n";
  });
  return 0;
}

// histo / stats need only the counts: for binary/sorted files the fixed-width records are split between a few threads
// over a read-only mapping (histo_main.cc runs -t threads over the file the same way); text files are read serially.
// make() builds one accumulator per thread, visit(acc, count) feeds it; the accumulators are returned for reduction.
template <typename Acc, typename Make, typename Visit>
static std::vector<Acc> scan_counts(const std::string& path, unsigned threads, Make make, Visit visit) {
  db_file f(path);
  std::vector<Acc> accs;
  if(f.header.format() != binary_dumper::format) {
    accs.push_back(make());
    for_each_record(f, [&](const mer_dna&, uint64_t v) { visit(accs[0], v); });
    return accs;
  }
  mapped_file map(path.c_str());
  const size_t off = f.header.offset(), kb = (f.header.key_len() + 7) / 8, vb = f.header.counter_len(), rec = kb + vb;
  const size_t n = map.length() > off ? (map.length() - off) / rec : 0;
  const unsigned char* base = (const unsigned char*)map.base() + off;
  const unsigned nt = (unsigned)std::max<size_t>(1, std::min<size_t>(threads, n >> 16));
  for(unsigned i = 0; i < nt; ++i) accs.push_back(make());
  auto work = [&](unsigned i) {
    const size_t a = n * i / nt, b = n * (i + 1) / nt;
    for(size_t r = a; r < b; ++r) {
      uint64_t v = 0;
      memcpy(&v, base + r * rec + kb, vb);
      visit(accs[i], v);
    }
  };
  std::vector<std::thread> th;
  for(unsigned i = 1; i < nt; ++i) th.emplace_back(work, i);
  work(0);
  for(auto& t : th) t.join();
  return accs;
}
static unsigned default_threads() { return std::max(1u, std::min(16u, std::thread::hardware_concurrency())); }

int histo_main(int argc, char* argv[]) {
  uint64_t low = 1, high = 10000, inc = 1;
  bool full = false;
  unsigned threads = default_threads();
  std::string output, db;
  ArgCursor a{argc, argv};
  for(; a.more(); ++a.i) {
    if(a.is("-l", "--low")) low = strtoull(a.value("-l", "--low").c_str(), 0, 10);
    else if(a.is("-h", "--high")) high = strtoull(a.value("-h", "--high").c_str(), 0, 10);
    else if(a.is("-i", "--increment")) inc = strtoull(a.value("-i", "--increment").c_str(), 0, 10);
    else if(a.is("-t", "--threads")) threads = std::max(1u, (unsigned)strtoul(a.value("-t", "--threads").c_str(), 0, 10));
    else if(a.cur() == "-f" || a.cur() == "--full") full = true;
    else if(a.is("-o", "--output")) output = a.value("-o", "--output");
    else db = positional(a);
  }
  if(db.empty()) die("Usage: jellyfish-amd histo [-l low] [-h high] [-i inc] [-f] [-o out] db:path");
  if(high < low) die("High count value must be >= to low count value");
  if(!inc) die("Increment must be > 0");
  std::ofstream fout;
  if(!output.empty()) { fout.open(output); if(!fout.good()) die("Error opening output file '" + output + "'"); }
  std::ostream& out = output.empty() ? std::cout : fout;
  const uint64_t base = inc >= low ? 0 : low - inc, ceil = high + inc;
  const uint64_t nb = (ceil + inc - base) / inc;
  typedef std::vector<uint64_t> hist_t;
  std::vector<hist_t> parts = scan_counts<hist_t>(db, threads, [&]() { return hist_t(nb, 0); }, [&](hist_t& h, uint64_t v) {
    if(v < base) ++h[0]; else if(v > ceil) ++h[nb - 1]; else ++h[(v - base) / inc];
  });
  hist_t histo(nb, 0);
  for(const auto& h : parts) for(uint64_t i = 0; i < nb; ++i) histo[i] += h[i];
  uint64_t col = base;
  for(uint64_t i = 0; i < nb; ++i, col += inc)
    if(histo[i] > 0 || full) out << col << " " << histo[i] << "\n";
  return 0;
}

// digest db: the content checksum documented with jfgpu_digest (include/jfgpu.h), from a file (host scan)
int digest_main(int argc, char* argv[]) {
  uint64_t lower = 0, upper = std::numeric_limits<uint64_t>::max();
  std::string db;
  ArgCursor a{argc, argv};
  for(; a.more(); ++a.i) {
    if(a.is("-L", "--lower-count")) lower = strtoull(a.value("-L", "--lower-count").c_str(), 0, 10);
    else if(a.is("-U", "--upper-count")) upper = strtoull(a.value("-U", "--upper-count").c_str(), 0, 10);
    else db = positional(a);
  }
  if(db.empty()) die("Usage: jellyfish-amd digest [-L l] [-U u] db:path");
  auto mix = [](uint64_t z) { z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); };
  db_file f(db);
  uint64_t n = 0, total = 0, sum = 0, x = 0;
  for_each_record(f, [&](const mer_dna& m, uint64_t v) {
    if(v < lower || v > upper) return;
    uint64_t h = 0x9E3779B97F4A7C15ull;
    for(unsigned w = 0; w < m.nb_words(); ++w) h = mix(h ^ m.word(w));
    h = mix(h ^ v);
    ++n; total += v; sum += h; x ^= h;
  });
  std::cout << "records " << n << "\ntotal " << total << "\nsum " << sum << "\nxor " << x << "\n";
  return 0;
}

int stats_main(int argc, char* argv[]) {
  uint64_t lower = 0, upper = std::numeric_limits<uint64_t>::max();
  std::string output, db;
  ArgCursor a{argc, argv};
  for(; a.more(); ++a.i) {
    if(a.is("-L", "--lower-count")) lower = strtoull(a.value("-L", "--lower-count").c_str(), 0, 10);
    else if(a.is("-U", "--upper-count")) upper = strtoull(a.value("-U", "--upper-count").c_str(), 0, 10);
    else if(a.is("-o", "--output")) output = a.value("-o", "--output");
    else db = positional(a);
  }
  if(db.empty()) die("Usage: jellyfish-amd stats [-L l] [-U u] [-o out] db:path");
  std::ofstream fout;
  if(!output.empty()) { fout.open(output); if(!fout.good()) die("Error opening output file '" + output + "'"); }
  std::ostream& out = output.empty() ? std::cout : fout;
  struct acc_t { uint64_t uniq = 0, distinct = 0, total = 0, max = 0; };
  std::vector<acc_t> parts = scan_counts<acc_t>(db, default_threads(), []() { return acc_t(); }, [&](acc_t& x, uint64_t v) {
    if(v < lower || v > upper) return;
    x.uniq += v == 1; x.total += v; x.max = std::max(x.max, v); ++x.distinct;
  });
  uint64_t uniq = 0, distinct = 0, total = 0, max = 0;
  for(const auto& x : parts) { uniq += x.uniq; distinct += x.distinct; total += x.total; max = std::max(max, x.max); }
  out << "Unique:    " << uniq << "\n" << "Distinct:  " << distinct << "\n"
      << "Total:     " << total << "\n" << "Max_count: " << max << "\n";
  return 0;
}

int query_main(int argc, char* argv[]) {
  std::string output, db;
  std::vector<std::string> mers, sequences;
  bool interactive = false;
  ArgCursor a{argc, argv};
  for(; a.more(); ++a.i) {
    if(a.is("-s", "--sequence")) sequences.push_back(a.value("-s", "--sequence"));
    else if(a.is("-o", "--output")) output = a.value("-o", "--output");
    else if(a.cur() == "-i" || a.cur() == "--interactive") interactive = true;
    else if(a.cur() == "-l" || a.cur() == "--load" || a.cur() == "-L" || a.cur() == "--no-load") {}
    else if(db.empty()) db = positional(a);
    else mers.push_back(positional(a));
  }
  if(db.empty()) die("Usage: jellyfish-amd query [-s file] [-i] [-o out] db:path [mers...]");
  std::ofstream fout;
  if(!output.empty()) { fout.open(output); if(!fout.good()) die("Error opening output file '" + output + "'"); }
  std::ostream& out = output.empty() ? std::cout : fout;
  std::ifstream in(db, std::ios::binary);
  file_header header(in);
  if(!in.good()) die("Failed to parse header of file '" + db + "'");
  mer_dna::k(header.key_len() / 2);
  if(header.format() != binary_dumper::format && header.format() != "bloomcounter")
    die("Unsupported format '" + header.format() + "'. Must be a bloom counter or binary list.");
  mapped_file map(db.c_str());
  // the two kinds of database answer the same question, "check(mer)" (query_main.cc:99-117)
  std::unique_ptr<binary_query> bin;
  std::unique_ptr<bloom_query> bloom;
  try {
    if(header.format() == "bloomcounter")
      bloom.reset(new bloom_query(map.base() + header.offset(), map.length() - header.offset(), header.size(), (unsigned)header.nb_hashes(),
                                  header.matrix(1), header.matrix(2)));
    else
      bin.reset(new binary_query(map.base() + header.offset(), header.key_len(), header.counter_len(), header.matrix(), header.size() - 1,
                                 map.length() - header.offset()));
  } catch(std::exception& e) { die(e.what()); }
  struct { binary_query* b; bloom_query* f; uint64_t check(const mer_dna& m) const { return b ? b->check(m) : f->check(m); } } bq{bin.get(), bloom.get()};
  const bool canonical = header.canonical();
  const unsigned k = mer_dna::k();
  // query_from_sequence (query_main.cc:44-51) answered by the device (batched GPU lookups): the file's records
  // go back into a table under the header's own matrix (jfgpu_add_key_vals), the k-mers of the -s files are looked up in
  // batches of a million (jfgpu_lookup: array::get_val_for_key over a batch).  Keys of one and two words, binary/sorted
  // databases; anything else -- or JFGPU_QUERY_HOST=1, or no device -- takes the host's binary search below.
  if(!sequences.empty() && bin && k <= 64 && !getenv("JFGPU_QUERY_HOST") && jfgpu_device_count() > 0) {
    jfgpu_params p; memset(&p, 0, sizeof p);
    p.k = k; p.canonical = canonical; p.size = header.size(); p.device = -1; p.out_counter_len = 8;
    const header_matrix hm = header.matrix();
    if(!hm.identity) p.matrix_columns = hm.columns.data();
    jfgpu_table* t = nullptr;
    if(jfgpu_create(&p, &t) != JFGPU_OK) {                   // (no room for the table on the device, ...: the host path answers)
      if(!getenv("JFGPU_QUIET")) std::cerr << "jellyfish-amd query: device table not created (" << jfgpu_last_error() << "): answering on the host\n";
      t = nullptr;
    }
    if(t) {
    const unsigned kw = (2 * k + 63) / 64, kb = (2 * k + 7) / 8, vb = header.counter_len();
    const size_t rec = kb + vb, n_rec = (map.length() - header.offset()) / rec;
    const unsigned char* body = reinterpret_cast<const unsigned char*>(map.base() + header.offset());
    const size_t kBatch = (size_t)1 << 20;
    std::vector<uint64_t> keys(kBatch * kw), vals(kBatch);
    for(size_t r0 = 0; r0 < n_rec; r0 += kBatch) {
      const size_t n = std::min(kBatch, n_rec - r0);
      std::fill(keys.begin(), keys.begin() + n * kw, 0); std::fill(vals.begin(), vals.begin() + n, 0);
      for(size_t i = 0; i < n; ++i) { memcpy(&keys[i * kw], body + (r0 + i) * rec, kb); memcpy(&vals[i], body + (r0 + i) * rec + kb, vb); }
      if(jfgpu_add_key_vals(t, keys.data(), vals.data(), n) != JFGPU_OK) { const std::string e = jfgpu_last_error(); jfgpu_destroy(t); die("query: " + e); }
    }
    std::vector<uint8_t> found(kBatch);
    size_t fill = 0;
    std::string text;
    auto answer = [&]() {
      if(!fill) return;
      if(jfgpu_lookup(t, keys.data(), fill, vals.data(), found.data()) != JFGPU_OK) { const std::string e = jfgpu_last_error(); jfgpu_destroy(t); die("query: " + e); }
      mer_dna m(k);
      text.clear();
      for(size_t i = 0; i < fill; ++i) {
        memcpy(m.data__(), &keys[i * kw], kw * sizeof(uint64_t));
        text += m.to_str(); text += ' '; text += std::to_string(vals[i]); text += '\n';
      }
      out.write(text.data(), text.size());
      fill = 0;
    };
    for(const auto& path : sequences) {
      sequence_parser parser(k);
      parser.parse_file(path.c_str(), [&](const char* buf, size_t n) {
        mer_dna m(k), rc(k);
        unsigned filled = 0;
        for(size_t i = 0; i < n; ++i) {
          const int code = mer_dna::code(buf[i]);
          if(code < 0) { filled = 0; continue; }
          m.shift_left(code); rc.shift_right(3 - code);
          if(++filled >= k) {
            filled = k;
            const mer_dna& q = (!canonical || m < rc) ? m : rc;
            for(unsigned w = 0; w < kw; ++w) keys[fill * kw + w] = q.word(w);
            if(++fill == kBatch) answer();
          }
        }
      });
    }
    answer();
    jfgpu_destroy(t);
    sequences.clear();
    }
  }
  for(const auto& path : sequences) {
    sequence_parser parser(k);
    parser.parse_file(path.c_str(), [&](const char* buf, size_t n) {
      mer_dna m(k), rc(k);
      unsigned filled = 0;
      for(size_t i = 0; i < n; ++i) {
        const int code = mer_dna::code(buf[i]);
        if(code < 0) { filled = 0; continue; }
        m.shift_left(code); rc.shift_right(3 - code);
        if(++filled >= k) {
          filled = k;
          const mer_dna& q = (!canonical || m < rc) ? m : rc;
          out << q << " " << bq.check(q) << "\n";
        }
      }
    });
  }
  auto one = [&](const std::string& s, bool show_mer) {
    try {
      mer_dna m(k);
      m = s;
      if(canonical) m.canonicalize();
      if(show_mer) out << m << " " << bq.check(m) << "\n";
      else out << bq.check(m) << std::endl;
    } catch(std::length_error&) { std::cerr << "Invalid mer '" << s << "'\n"; }
  };
  for(const auto& s : mers) one(s, true);
  if(interactive) { std::string line; while(std::getline(std::cin, line)) one(line, false); }
  return 0;
}

// ---------------------------------------------------------------- mem  (sub_commands/mem_main.cc)
// Device memory of the table for a size hint, or the largest size hint that fits a memory budget.
static std::string add_suffix(uint64_t x, uint64_t unit) {
  static const char suffixes[] = "kMGTPE";
  int i = 0;
  while(x >= unit && i <= 5) { x /= unit; ++i; }
  return std::to_string(x) + (i > 0 ? std::string(1, suffixes[i - 1]) : std::string());
}
int mem_main(int argc, char* argv[]) {
  unsigned mer_len = 0; uint64_t size = 0, mem = 0; bool size_given = false, mem_given = false;
  ArgCursor a{argc, argv};
  for(; a.more(); ++a.i) {
    if(a.is("-m", "--mer-len")) mer_len = (unsigned)strtoul(a.value("-m", "--mer-len").c_str(), 0, 10);
    else if(a.is("-s", "--size")) { size = parse_suffix(a.value("-s", "--size"), "-s"); size_given = true; }
    else if(a.is("", "--mem")) { mem = parse_suffix(a.value("", "--mem"), "--mem"); mem_given = true; }
    else if(a.is("-c", "--counter-len")) (void)a.value("-c", "--counter-len");
    else if(a.is("-p", "--reprobes")) (void)a.value("-p", "--reprobes");
    else die("Unknown option '" + a.cur() + "'");
  }
  if(!mer_len) die("Error: mandatory switch missing: -m, --mer-len");
  if(size_given == mem_given) die("Error: exactly one of -s, --size and --mem is required");
  uint64_t slots = 0, bytes = 0;
  if(size_given) {
    if(jfgpu_table_bytes(mer_len, size, &slots, &bytes)) die(jfgpu_last_error());
    std::cout << bytes << " (" << add_suffix(bytes, 1024) << ")\n";
  } else {
    uint64_t best = 0;
    for(unsigned l = 1; l < 63; ++l) {
      if(jfgpu_table_bytes(mer_len, (uint64_t)1 << l, &slots, &bytes)) die(jfgpu_last_error());
      if(bytes <= mem && slots == ((uint64_t)1 << l)) best = slots;
    }
    std::cout << best << " (" << add_suffix(best, 1000) << ")\n";
  }
  return 0;
}

int info_main(int argc, char* argv[]) {
  bool json = false, skip = false, cmd = false;
  std::string db;
  for(int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    if(a == "-j" || a == "--json") json = true; else if(a == "-s" || a == "--skip") skip = true;
    else if(a == "-c" || a == "--cmd") cmd = true;
    else if(a.size() > 1 && a[0] == '-') die("Unknown option '" + a + "'");
    else db = a;
  }
  if(db.empty()) die("Usage: jellyfish-amd info [-j] [-c] [-s] db:path");
  std::ifstream is(db, std::ios::binary);
  file_header header;
  if(!header.read(is)) die("Failed to parse header of file '" + db + "'");
  if(skip) { std::cout << is.rdbuf(); return 0; }
  if(json) { std::cout << header.root().dump() << "\n"; return 0; }
  if(cmd) { for(const auto& s : header.cmdline()) std::cout << s << " "; std::cout << "\n"; return 0; }
  std::cout << "command: "; for(const auto& s : header.cmdline()) std::cout << s << " "; std::cout << "\n";
  std::cout << "where: " << header["hostname"] << ":" << header["pwd"] << "\n"
            << "when: " << header["time"] << "\n"
            << "canonical: " << (header.canonical() ? "yes" : "no") << "\n";
  return 0;
}

}  // namespace

int main(int argc, char* argv[]) {
  const char* usage =
      "Usage: jellyfish-amd <cmd> [options] arg...\n"
      "Where <cmd> is one of: count, bc, merge, stats, histo, dump, query, mem, info.\n"
      "Options:\n  --version        Display version\n  --help           Display this message\n";
  if(argc < 2) { std::cerr << "Too few arguments\n" << usage; return 1; }
  const std::string cmd = argv[1];
  if(cmd == "--version" || cmd == "-V") { std::cout << "jellyfish-amd 0.1 (MI355X engine for jellyfish 2.3.1 files)\n"; return 0; }
  if(cmd == "--help" || cmd == "-h") { std::cout << usage; return 0; }
  try {
    if(cmd == "count") return count_main(argc - 1, argv + 1);
    if(cmd == "bc") return bc_main(argc - 1, argv + 1);
    if(cmd == "dump") return dump_main(argc - 1, argv + 1);
    if(cmd == "histo") return histo_main(argc - 1, argv + 1);
    if(cmd == "stats") return stats_main(argc - 1, argv + 1);
    if(cmd == "digest") return digest_main(argc - 1, argv + 1);
    if(cmd == "query") return query_main(argc - 1, argv + 1);
    if(cmd == "merge") return merge_main(argc - 1, argv + 1);
    if(cmd == "mem") return mem_main(argc - 1, argv + 1);
    if(cmd == "info") return info_main(argc - 1, argv + 1);
  } catch(std::exception& e) { die(e.what()); }
  std::cerr << "Unknown command '" << cmd << "'\n" << usage;
  return 1;
}
