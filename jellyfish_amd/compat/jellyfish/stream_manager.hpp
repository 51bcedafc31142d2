// jellyfish/stream_manager.hpp (compat): the list of input paths (include/jellyfish/stream_manager.hpp:31-120).
// Only the file part: the engine's readers open the paths themselves.
#pragma once
#include <string>
#include <vector>
namespace jellyfish {
template <typename PathIterator>
class stream_manager {
public:
  stream_manager(PathIterator paths_begin, PathIterator paths_end, int concurrent_files = 1) : concurrent_(concurrent_files) {
    for(PathIterator it = paths_begin; it != paths_end; ++it) paths_.push_back(std::string(*it));
  }
  stream_manager(PathIterator paths_begin, PathIterator paths_end, PathIterator, PathIterator, int concurrent_files = 1)
      : stream_manager(paths_begin, paths_end, concurrent_files) {}
  int concurrent_files() const { return concurrent_; }
  int nb_streams() const { return concurrent_; }
  const std::vector<std::string>& paths() const { return paths_; }
private:
  std::vector<std::string> paths_;
  int concurrent_;
};
}  // namespace jellyfish
