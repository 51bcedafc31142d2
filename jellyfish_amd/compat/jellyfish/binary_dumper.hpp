// jellyfish/binary_dumper.hpp (compat): the read side of binary/sorted files -- jellyfish::binary_reader<Key, Val>,
// binary_query_base<Key, Val> (include/jellyfish/binary_dumper.hpp:83-213) -- and the format name of the writer, over the
// engine's own reader / query classes (jellyfish_amd/dumpers.hpp; same record layout, same interpolation search).
#pragma once
#include <jellyfish_amd/dumpers.hpp>
#include <jellyfish/file_header.hpp>
#include <jellyfish/mer_dna.hpp>
namespace jellyfish {
template <typename Key, typename Val>
class binary_reader : public jellyfish_amd::binary_reader {
public:
  binary_reader(std::istream& is, file_header* header) : jellyfish_amd::binary_reader(is, header) {}
};
template <typename Key, typename Val>
class binary_query_base : public jellyfish_amd::binary_query {
public:
  binary_query_base(const char* data, unsigned key_len, unsigned val_len, const RectangularBinaryMatrix& m, size_t mask, size_t size)
      : jellyfish_amd::binary_query(data, key_len, val_len, m, mask, size) {}
};
template <typename Storage>
struct binary_dumper_format { static constexpr const char* format = "binary/sorted"; };
using jellyfish_amd::mapped_file;
}  // namespace jellyfish
