// jellyfish/jellyfish.hpp (compat): the global names of include/jellyfish/jellyfish.hpp:18-27 a client program expects.
#pragma once
#include <jellyfish/mer_dna.hpp>
#include <jellyfish/hash_counter.hpp>
#include <jellyfish/file_header.hpp>
#include <jellyfish/text_dumper.hpp>
#include <jellyfish/binary_dumper.hpp>
typedef jellyfish::cooperative::hash_counter<jellyfish::mer_dna> mer_hash;
typedef mer_hash::array mer_array;
typedef jellyfish::text_reader<jellyfish::mer_dna, uint64_t> text_reader;
typedef jellyfish::binary_reader<jellyfish::mer_dna, uint64_t> binary_reader;
typedef jellyfish::binary_query_base<jellyfish::mer_dna, uint64_t> binary_query;
struct binary_dumper { static constexpr const char* format = "binary/sorted"; };    // (writing goes through the engine: jellyfish_amd::binary_dumper)
struct text_dumper { static constexpr const char* format = "text/sorted"; };
