// The hash that ties a run-time-compiled kernel instance to the sources and flags of the library that launches it (beat_ode_jit.hip):
// every header of csrc/ (sorted by name) and include/beat_hip.h, FNV-1a.  Used twice: by tools/jit_hash.cpp at BUILD time (csrc/Makefile
// writes the value into beat_build_hash.h, together with the flags) and by the library at RUN time on the sources it finds beside
// itself -- unequal: no run-time compilation (the run-time-index kernel runs instead).  Host code only.
#pragma once
#include <dirent.h>

#include <algorithm>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

namespace beat_jit_hash {
inline unsigned long long fnv(const std::string& s, unsigned long long h = 1469598103934665603ull) {
  for (unsigned char c : s) {
    h ^= c;
    h *= 1099511628211ull;
  }
  return h;
}
inline std::string slurp(const std::string& p) {
  std::ifstream f(p, std::ios::binary);
  std::ostringstream o;
  o << f.rdbuf();
  return o.str();
}
// csrc = the directory of the kernel headers; include/beat_hip.h is found two levels up
inline unsigned long long beat_jit_source_hash(const std::string& csrc) {
  std::vector<std::string> names;
  if (DIR* d = ::opendir(csrc.c_str())) {
    while (dirent* de = ::readdir(d)) {
      const std::string n = de->d_name;
      if (n.size() > 2 && n.compare(n.size() - 2, 2, ".h") == 0 && n != "beat_build_hash.h") names.push_back(n);
    }
    ::closedir(d);
  }
  std::sort(names.begin(), names.end());
  unsigned long long h = fnv("beat-jit-sources");
  for (const std::string& n : names) h = fnv(slurp(csrc + "/" + n), fnv(n, h));
  return fnv(slurp(csrc + "/../../include/beat_hip.h"), h);
}
}  // namespace beat_jit_hash
