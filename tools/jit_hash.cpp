// Build-time half of the run-time compiler's source check (csrc/beat_jit_hash.h): prints the header csrc/Makefile includes into the
// library.   usage: jit_hash <csrc directory>
#include <cstdio>

#include "../fenicsx-beat_amd/csrc/beat_jit_hash.h"

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  std::printf("// written by csrc/Makefile (tools/jit_hash.cpp): the kernel headers this library was compiled from\n"
              "#pragma once\n#define BEAT_BUILD_SRC_HASH 0x%016llxull\n",
              beat_jit_hash::beat_jit_source_hash(argv[1]));
  return 0;
}
