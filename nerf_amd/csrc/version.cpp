// version.cpp -- nerfhip_version(): release + the hash of the kernel sources this library was built from (Makefile: SRC_HASH,
// sha256 over *.hip *.h *.inc of this directory).  A measurement file that records the string can be matched to a library.
#ifndef NERFHIP_SRC_HASH
#define NERFHIP_SRC_HASH "unknown"
#endif
extern "C" const char* nerfhip_version(void) {
  return "nerfhip 0.3 (gfx950; fp32, bf16 and split-f16 MFMA) src " NERFHIP_SRC_HASH;
}
