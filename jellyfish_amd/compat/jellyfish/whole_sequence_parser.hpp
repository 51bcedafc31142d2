// jellyfish/whole_sequence_parser.hpp (compat): whole FASTA / FASTQ records (header, sequence, qualities) in batches
// (include/jellyfish/whole_sequence_parser.hpp:19-180).  Host side only -- it feeds per-record clients such as
// examples/query_per_sequence, not the counting path; single consumer (the reference's pool hands batches to several).
#pragma once
#include <fstream>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>
#include <jellyfish/stream_manager.hpp>
namespace jellyfish {
struct header_sequence_qual { std::string header, seq, qual; };
struct sequence_list { size_t nb_filled = 0; std::vector<header_sequence_qual> data; };

template <typename StreamIterator>
class whole_sequence_parser {
public:
  // size: number of batches in flight (unused here), nb_sequences: records per batch, max_producers: unused
  whole_sequence_parser(uint32_t, uint32_t nb_sequences, uint32_t, StreamIterator& streams) : streams_(streams), per_batch_(nb_sequences ? nb_sequences : 1) {}

  class job {
  public:
    explicit job(whole_sequence_parser& p) { p.fill(list_); }
    bool is_empty() const { return list_.nb_filled == 0; }
    sequence_list* operator->() { return &list_; }
    sequence_list& operator*() { return list_; }
    void next() {}
  private:
    sequence_list list_;
  };

private:
  friend class job;
  StreamIterator& streams_;
  size_t per_batch_, file_ = 0;
  std::unique_ptr<std::ifstream> in_;
  enum { NONE, FASTA, FASTQ } type_ = NONE;
  std::string pending_;                       // the header line read while looking for the end of a FASTA record

  bool open_next() {
    while(file_ < streams_.paths().size()) {
      in_.reset(new std::ifstream(streams_.paths()[file_++]));
      if(!in_->good()) throw std::runtime_error("Can't open file '" + streams_.paths()[file_ - 1] + "'");
      const int c = in_->peek();
      if(c == '>') { type_ = FASTA; pending_.clear(); return true; }
      if(c == '@') { type_ = FASTQ; return true; }
      if(c == std::char_traits<char>::eof()) continue;        // empty file
      throw std::runtime_error("Unsupported format");
    }
    in_.reset();
    return false;
  }
  bool read_record(header_sequence_qual& r) {
    r.header.clear(); r.seq.clear(); r.qual.clear();
    std::string line;
    for(;;) {
      if(!in_ && !open_next()) return false;
      if(type_ == FASTA) {
        if(pending_.empty()) { if(!std::getline(*in_, pending_)) { in_.reset(); continue; } }
        r.header = pending_.substr(1);
        pending_.clear();
        while(in_->peek() != '>' && std::getline(*in_, line)) { if(!line.empty() && line.back() == '\r') line.pop_back(); r.seq += line; }
        if(in_->peek() == '>') std::getline(*in_, pending_);
        else in_.reset();
        return true;
      }
      if(!std::getline(*in_, line)) { in_.reset(); continue; }
      r.header = line.substr(1);
      while(in_->peek() != '+' && std::getline(*in_, line)) { if(!line.empty() && line.back() == '\r') line.pop_back(); r.seq += line; }
      if(!std::getline(*in_, line)) throw std::runtime_error("Truncated fastq file");       // the '+' line
      while(r.qual.size() < r.seq.size() && std::getline(*in_, line)) { if(!line.empty() && line.back() == '\r') line.pop_back(); r.qual += line; }
      if(r.qual.size() != r.seq.size()) throw std::runtime_error("Invalid fastq file: wrong number of quals");
      if(in_->peek() == std::char_traits<char>::eof()) in_.reset();
      return true;
    }
  }
  void fill(sequence_list& l) {
    l.data.resize(per_batch_);
    l.nb_filled = 0;
    while(l.nb_filled < per_batch_ && read_record(l.data[l.nb_filled])) ++l.nb_filled;
  }
};
}  // namespace jellyfish
