// jellyfish/rectangular_binary_matrix.hpp (compat): the GF(2) matrix of a file header as client programs hold it
// (include/jellyfish/rectangular_binary_matrix.hpp:29-150: r x c, times(mer), comparison).  Construction of NEW matrices
// (randomize / pseudo-inverse) is the engine's business (csrc/gf2_matrix.hpp); this is the read side.
#pragma once
#include <jellyfish_amd/file_header.hpp>
#include <jellyfish/mer_dna.hpp>
namespace jellyfish {
class RectangularBinaryMatrix : public jellyfish_amd::header_matrix {
public:
  RectangularBinaryMatrix() {}
  RectangularBinaryMatrix(unsigned rows, unsigned cols) { jellyfish_amd::header_matrix::r = rows; jellyfish_amd::header_matrix::c = cols; columns.assign(cols, 0); }
  RectangularBinaryMatrix(const jellyfish_amd::header_matrix& m) : jellyfish_amd::header_matrix(m) {}
  unsigned r() const { return jellyfish_amd::header_matrix::r; }
  unsigned c() const { return jellyfish_amd::header_matrix::c; }
  bool is_zero() const { for(uint64_t x : columns) if(x) return false; return !identity; }
  uint64_t times(const mer_dna& m) const { return jellyfish_amd::header_matrix::times(m.data()); }
  uint64_t times(const uint64_t* words) const { return jellyfish_amd::header_matrix::times(words); }
  bool operator==(const RectangularBinaryMatrix& o) const {
    return jellyfish_amd::header_matrix::r == o.jellyfish_amd::header_matrix::r && jellyfish_amd::header_matrix::c == o.jellyfish_amd::header_matrix::c &&
           identity == o.identity && columns == o.columns;
  }
  bool operator!=(const RectangularBinaryMatrix& o) const { return !(*this == o); }
};
}  // namespace jellyfish
