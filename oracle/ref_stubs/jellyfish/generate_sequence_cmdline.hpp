// Stand-in for the yaggo-generated option struct of the reference's
// jellyfish/generate_sequence.cc (yaggo is not installed here).  Field names
// follow jellyfish/generate_sequence_cmdline.yaggo; only what main() reads.
// usage: ref_generate_sequence -s SEED [-o PREFIX] [-r READLEN] [-q] [-v] LENGTH...
#ifndef ORACLE_GENERATE_SEQUENCE_CMDLINE_HPP
#define ORACLE_GENERATE_SEQUENCE_CMDLINE_HPP
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <vector>
struct generate_sequence_args {
  long                  seed_arg;
  std::vector<uint32_t> mer_arg;
  const char*           output_arg;
  bool                  fastq_flag, verbose_flag, read_length_given;
  uint32_t              read_length_arg;
  std::vector<uint64_t> length_arg;
  generate_sequence_args() : seed_arg(0), output_arg("output"), fastq_flag(false),
                             verbose_flag(false), read_length_given(false), read_length_arg(0) {}
  void parse(int argc, char* argv[]) {
    bool seed_given = false;
    for(int i = 1; i < argc; ++i) {
      const char* a = argv[i];
      if(!strcmp(a, "-s") && i + 1 < argc)      { seed_arg = atol(argv[++i]); seed_given = true; }
      else if(!strcmp(a, "-o") && i + 1 < argc) { output_arg = argv[++i]; }
      else if(!strcmp(a, "-r") && i + 1 < argc) { read_length_arg = strtoul(argv[++i], 0, 10); read_length_given = true; }
      else if(!strcmp(a, "-m") && i + 1 < argc) { mer_arg.push_back(strtoul(argv[++i], 0, 10)); }
      else if(!strcmp(a, "-q"))                 { fastq_flag = true; }
      else if(!strcmp(a, "-v"))                 { verbose_flag = true; }
      else                                      { length_arg.push_back(strtoull(a, 0, 10)); }
    }
    if(!seed_given || length_arg.empty()) {
      fprintf(stderr, "usage: %s -s SEED [-o PREFIX] [-r READLEN] [-q] LENGTH...\n", argv[0]);
      exit(1);
    }
  }
};
#endif
