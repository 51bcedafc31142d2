// jellyfish/file_header.hpp (compat): jellyfish::file_header as client programs read it (include/jellyfish/file_header.hpp:18-109,
// generic_file_header.hpp) over the engine's own header class (same JSON, same 9-digit length prefix).
#pragma once
#include <jellyfish_amd/file_header.hpp>
#include <jellyfish/rectangular_binary_matrix.hpp>
namespace jellyfish {
class file_header : public jellyfish_amd::file_header {
public:
  file_header() {}
  explicit file_header(std::istream& is) : jellyfish_amd::file_header(is) {}
  RectangularBinaryMatrix matrix(int i = 1) const { return RectangularBinaryMatrix(jellyfish_amd::file_header::matrix(i)); }
  using jellyfish_amd::file_header::get_reprobes;
  void get_reprobes(size_t* r) const { const std::vector<size_t> v = jellyfish_amd::file_header::get_reprobes(); for(size_t i = 0; i < v.size(); ++i) r[i] = v[i]; }      // file_header.hpp:73-77
};
}  // namespace jellyfish
