// tests/host/parser_emu.cc -- the host sequence reader (jellyfish_amd/include/jellyfish_amd/sequence_parser.hpp)
// as a filter: file in, the concatenation of its contract buffers (seams removed) out.  Used by tests/test_host.py to
// compare it with the oracle's restatement of mer_overlap_sequence_parser on arbitrary layouts, without a GPU.
// usage: parser_emu <k> <buffer bytes> <file>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>

#include <jellyfish_amd/sequence_parser.hpp>

int main(int argc, char* argv[]) {
  if(argc != 4) return 2;
  const unsigned k = (unsigned)atoi(argv[1]);
  const size_t buf = (size_t)atoll(argv[2]);
  jellyfish_amd::sequence_parser parser(k, buf);
  std::string out;
  bool first = true;
  try {
    parser.parse_file(argv[3], [&](const char* p, size_t n) {
      // every buffer but the first of a file starts with the previous buffer's last k-1 characters
      const size_t skip = first ? 0 : std::min<size_t>(k - 1, n);
      out.append(p + skip, n - skip);
      first = false;
    });
  } catch(std::exception& e) {
    fprintf(stderr, "%s\n", e.what());
    return 1;
  }
  fwrite(out.data(), 1, out.size(), stdout);
  return 0;
}
