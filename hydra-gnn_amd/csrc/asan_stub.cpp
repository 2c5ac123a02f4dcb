// Host-only stand-in for the two symbols htree.cpp takes from runtime.hip, so that `make asan` can build the H-tree construction
// with g++ -fsanitize=address,undefined (GPU sanitizers are not available on this pool; the CPU build is the instrumented one).
#include <cstring>

#include "../../include/hydra_mp.h"

namespace hmp {
char* err_buf() {
  static thread_local char buf[512];
  return buf;
}
}  // namespace hmp

extern "C" const char* hmp_last_error(void) { return hmp::err_buf(); }
extern "C" int hmp_abi_version(void) { return HMP_ABI_VERSION; }
