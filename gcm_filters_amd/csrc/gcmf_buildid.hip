// The identity of the sources this binary was built from (gcm_filters_amd/_build.py: sha256 over csrc/*.{hip,hpp,h},
// include/gcmf.h and the compiler flags).  _lib.load() compares it with the sources next to the binary and refuses
// (or rebuilds) a stale libgcmf.so.  The marker string is what _build.binary_build_id() finds without dlopen.
#include "gcmf.h"

#ifndef GCMF_BUILD_ID
#error "gcmf_buildid.hip is compiled by _build.py with -DGCMF_BUILD_ID=\"<sha256>\""
#endif

extern "C" {
__attribute__((used)) const char gcmf_build_id_marker[] = "GCMF_BUILD_ID=" GCMF_BUILD_ID;
const char* gcmf_build_id(void) { return gcmf_build_id_marker + 14; }
}
