// jellyfish_amd/include/jellyfish_amd/file_header.hpp
//
// The JSON file header of every Jellyfish output file, byte-compatible with
// include/jellyfish/generic_file_header.hpp:88-143 and file_header.hpp:18-109:
//   bytes 0-8  : header length as 9 zero-padded decimal digits
//   then       : compact JSON, NUL-padded so that (9 + hlen) % alignment == 0
// and the same keys (format, key_len, counter_len, size, matrix1{r,c,identity,
// columns[]}, canonical, max_reprobe, reprobes[], val_len, cmdline[], hostname,
// pwd, time, exe_path, alignment; Bloom files: matrix2, nb_hashes, fpr).
#pragma once
#include <unistd.h>
#include <sys/utsname.h>
#include <climits>
#include <ctime>
#include <istream>
#include <ostream>
#include <string>
#include <vector>

#include "json_min.hpp"

namespace jellyfish_amd {

// r x c GF(2) matrix as the header stores it: c column words, columns[c-1-j] is the image
// of key bit j (rectangular_binary_matrix.hpp:223-261).
struct header_matrix {
  unsigned r = 0, c = 0;
  bool identity = false;
  std::vector<uint64_t> columns;

  uint64_t times(const uint64_t* key_words) const {
    if(identity) return key_words[0] & (r >= 64 ? ~(uint64_t)0 : (((uint64_t)1 << r) - 1));
    uint64_t res = 0;
    for(unsigned j = 0; j < c; ++j)
      if((key_words[j / 64] >> (j % 64)) & 1) res ^= columns[c - 1 - j];
    return res;
  }
};

class file_header {
public:
  static const int MAX_HEADER_DIGITS = 9;

  file_header() { root_["alignment"] = Json(8u); }
  explicit file_header(std::istream& is) { root_["alignment"] = Json(8u); read(is); }

  // generic_file_header.hpp:88-111
  void write(std::ostream& os) {
    std::string header = root_.dump();
    const int align = alignment();
    size_t hlen = header.size();
    size_t pad = 0;
    if(align > 0) {
      const size_t rem = (MAX_HEADER_DIGITS + header.size()) % align;
      if(rem) pad = align - rem;
      hlen += pad;
    }
    char len[16];
    snprintf(len, sizeof len, "%09zu", hlen);
    os.write(len, MAX_HEADER_DIGITS);
    os.write(header.data(), header.size());
    offset_ = MAX_HEADER_DIGITS + hlen;
    for(size_t i = 0; i < pad; ++i) os.put('\0');
  }

  // generic_file_header.hpp:119-143
  bool read(std::istream& is) {
    std::string len;
    for(int i = 0; i < MAX_HEADER_DIGITS && isdigit(is.peek()); ++i) len += (char)is.get();
    if(is.peek() != '{') return false;
    const unsigned long hlen = strtoul(len.c_str(), nullptr, 10);
    if(hlen < 2) return false;
    offset_ = MAX_HEADER_DIGITS + hlen;
    std::string buf(hlen, '\0');
    is.read(&buf[0], hlen);
    if(!is.good()) return false;
    size_t end = hlen;
    while(end > 0 && buf[end - 1] == '\0') --end;
    try { root_ = Json::parse(buf.data(), buf.data() + end); } catch(std::exception&) { return false; }
    return true;
  }

  const Json& root() const { return root_; }
  Json& root() { return root_; }
  size_t offset() const { return offset_; }
  int alignment() const { return (int)root_.get("alignment").as_uint64(0); }

  // generic_file_header.hpp:147-213 (SOURCE_DATE_EPOCH makes the provenance reproducible)
  void fill_standard() {
    const bool repro = getenv("SOURCE_DATE_EPOCH") != nullptr;
    struct utsname u;
    root_["hostname"] = Json(repro ? "hostname" : (uname(&u) == 0 ? u.nodename : ""));
    char path[PATH_MAX + 1];
    root_["pwd"] = Json(repro ? "." : (getcwd(path, sizeof path) ? path : ""));
    time_t t = time(nullptr);
    std::string ts;
    if(repro) { t = (time_t)strtoll(getenv("SOURCE_DATE_EPOCH"), nullptr, 10); ts = asctime(gmtime(&t)); }
    else ts = ctime(&t);
    while(!ts.empty() && isspace((unsigned char)ts.back())) ts.pop_back();
    root_["time"] = Json(ts);
    ssize_t n = readlink("/proc/self/exe", path, PATH_MAX);
    root_["exe_path"] = Json(n > 0 ? std::string(path, n) : std::string());
  }
  void set_cmdline(int argc, char* argv[]) {
    Json a; a.set_array();
    for(int i = 0; i < argc; ++i) a.append(Json(argv[i]));
    root_["cmdline"] = a;
  }
  std::vector<std::string> cmdline() const {
    std::vector<std::string> res;
    const Json& a = root_.get("cmdline");
    for(size_t i = 0; i < a.size(); ++i) res.push_back(a.at(i).as_string());
    return res;
  }
  std::string operator[](const std::string& key) const { return root_.get(key).as_string(); }

  // file_header.hpp:35-64
  header_matrix matrix(int i = 1) const {
    const Json& m = root_.get("matrix" + std::to_string(i));
    header_matrix res;
    res.r = (unsigned)m.get("r").as_uint64();
    res.c = (unsigned)m.get("c").as_uint64();
    res.identity = m.get("identity").as_bool();
    if(!res.identity) {
      res.columns.assign(res.c, 0);
      for(unsigned j = 0; j < res.c; ++j) res.columns[j] = m.get("columns").at(j).as_uint64();
    }
    return res;
  }
  void matrix(const header_matrix& m, int i = 1) {
    Json j;
    j["r"] = Json(m.r);
    j["c"] = Json(m.c);
    j["identity"] = Json(m.identity);
    if(!m.identity) {
      Json cols; cols.set_array();
      for(unsigned x = 0; x < m.c; ++x) cols.append(Json((unsigned long long)m.columns[x]));
      j["columns"] = cols;
    }
    root_["matrix" + std::to_string(i)] = j;
  }

  size_t size() const { return root_.get("size").as_uint64(); }
  void size(size_t s) { root_["size"] = Json((unsigned long long)s); }
  unsigned key_len() const { return (unsigned)root_.get("key_len").as_uint64(); }
  void key_len(unsigned k) { root_["key_len"] = Json(k); }
  unsigned val_len() const { return (unsigned)root_.get("val_len").as_uint64(); }
  void val_len(unsigned k) { root_["val_len"] = Json(k); }
  unsigned max_reprobe() const { return (unsigned)root_.get("max_reprobe").as_uint64(); }
  void max_reprobe(unsigned m) { root_["max_reprobe"] = Json(m); }
  size_t max_reprobe_offset() const { return root_.get("reprobes").at(max_reprobe()).as_uint64(); }
  std::vector<size_t> get_reprobes() const {
    std::vector<size_t> r;
    const Json& a = root_.get("reprobes");
    for(size_t i = 0; i < a.size(); ++i) r.push_back((size_t)a.at(i).as_uint64());
    return r;
  }
  void set_reprobes(const std::vector<size_t>& r) {
    Json a; a.set_array();
    for(unsigned i = 0; i <= max_reprobe() && i < r.size(); ++i) a.append(Json((unsigned long long)r[i]));
    root_["reprobes"] = a;
  }
  double fpr() const { return root_.get("fpr").as_double(); }
  void fpr(double f) { root_["fpr"] = Json(f); }
  unsigned long nb_hashes() const { return root_.get("nb_hashes").as_uint64(); }
  void nb_hashes(unsigned long n) { root_["nb_hashes"] = Json((unsigned long long)n); }
  bool canonical() const { return root_.get("canonical").as_bool(false); }
  void canonical(bool v) { root_["canonical"] = Json(v); }
  unsigned counter_len() const { return (unsigned)root_.get("counter_len").as_uint64(); }
  void counter_len(unsigned l) { root_["counter_len"] = Json(l); }
  std::string format() const { return root_.get("format").as_string(); }
  void format(const std::string& s) { root_["format"] = Json(s); }

private:
  Json root_;
  size_t offset_ = 0;
};

}  // namespace jellyfish_amd
