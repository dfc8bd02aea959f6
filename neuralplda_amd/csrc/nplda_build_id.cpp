// nplda_build_id.cpp — which sources this libnplda_hip.so was built from.
// neuralplda_amd/build.py hashes csrc/* and include/nplda_hip.h (sha256, first 16 hex digits) and passes the digest as
// NPLDA_SRC_SHA; a box without hipcc uses whatever library travelled with the tree, and this makes a stale one visible
// (bench.py reports it next to the digest of the sources it finds: `lib.csrc_sha` / `lib.tree_sha` / `lib.stale`).
#include "../../include/nplda_hip.h"

#ifndef NPLDA_SRC_SHA
#define NPLDA_SRC_SHA "unknown"
#endif

extern "C" const char* nplda_build_id(void) { return NPLDA_SRC_SHA; }
