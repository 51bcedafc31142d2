// jellyfish/mer_dna.hpp of the compatibility include directory (-Ijellyfish_amd/compat): programs written against the
// reference's C++ API (include/jellyfish/*.hpp, namespace jellyfish) compile against the MI355X engine with their
// sources unchanged.  jellyfish::mer_dna is the engine's host-side k-mer type (same word layout: mer_dna.hpp:143-155).
#pragma once
#include <jellyfish_amd/mer_dna.hpp>
namespace jellyfish {
using jellyfish_amd::mer_dna;
}
