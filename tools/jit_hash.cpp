// Build-time half of the run-time compiler's source check (csrc/beat_jit_hash.h): prints the header csrc/Makefile includes into the
// library.   usage: jit_hash <csrc directory> [the Makefile's EXTRA flags]
#include <cstdio>
#include <string>

#include "../fenicsx-beat_amd/csrc/beat_jit_hash.h"

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  std::printf("// written by csrc/Makefile (tools/jit_hash.cpp): the kernel headers this library was compiled from\n"
              "#pragma once\n#define BEAT_BUILD_SRC_HASH 0x%016llxull\n",
              beat_jit_hash::beat_jit_source_hash(argv[1]));
  // the Makefile's EXTRA flags (variant libraries for A/B runs: -DBEAT_ODE_NT=3 ...): the run-time compiler passes them on and
  // hashes them into its cache key, so that an instance compiled at run time is built like the library that launches it
  std::string extra = argc > 2 ? argv[2] : "";
  std::string esc;
  for (char c : extra) {
    if (c == '"' || c == '\\') esc += '\\';
    esc += c;
  }
  std::printf("#define BEAT_BUILD_EXTRA_FLAGS \"%s\"\n", esc.c_str());
  return 0;
}
