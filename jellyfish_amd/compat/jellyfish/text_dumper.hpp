// jellyfish/text_dumper.hpp (compat): jellyfish::text_reader<Key, Val> (include/jellyfish/text_dumper.hpp:60-95) and the
// format name, over the engine's text reader.
#pragma once
#include <jellyfish_amd/dumpers.hpp>
#include <jellyfish/file_header.hpp>
namespace jellyfish {
template <typename Key, typename Val>
class text_reader : public jellyfish_amd::text_reader {
public:
  text_reader(std::istream& is, file_header* header) : jellyfish_amd::text_reader(is, header) {}
  size_t pos() const { return 0; }       // (text/sorted files are in key order of the dump; the heap falls back on the key)
};
}  // namespace jellyfish
