// Library identification (include/nvsf_hip.h: nvsf_version) and the test-only variant table (nvsf_test_variant).
#include "common.h"
#include <string.h>
NVSF_API const char* nvsf_version(void) { return "nvsf_hip 0.1.0 gfx950"; }

// Identity of the sources this object was built from (build.py passes both digests on the command line of THIS translation unit
// only; build.embedded_digest() finds the marker in the file's bytes, nvsf_build_digest() returns it from the mapped library).
#ifndef NVSF_CSRC_DIGEST_ALL
#define NVSF_CSRC_DIGEST_ALL "unknown_________"
#endif
#ifndef NVSF_CSRC_DIGEST_RENDER
#define NVSF_CSRC_DIGEST_RENDER "unknown_________"
#endif
#define NVSF_MARK_ALL "NVSF_CSRC_DIGEST_ALL="
#define NVSF_MARK_RENDER "NVSF_CSRC_DIGEST_RENDER="
static const char g_digest_all[] = NVSF_MARK_ALL NVSF_CSRC_DIGEST_ALL;
static const char g_digest_render[] = NVSF_MARK_RENDER NVSF_CSRC_DIGEST_RENDER;
NVSF_API const char* nvsf_build_digest(int which) {
    return which == 1 ? g_digest_render + sizeof(NVSF_MARK_RENDER) - 1 : g_digest_all + sizeof(NVSF_MARK_ALL) - 1;
}

static int g_variant[kVarCount] = {};
__attribute__((visibility("hidden"))) int nvsf_variant(int key) { return key >= 0 && key < kVarCount ? g_variant[key] : 0; }

__attribute__((visibility("hidden"))) int nvsf_cu_count() {
    static const int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        return v;
    }();
    return n;
}

NVSF_API int nvsf_test_variant(const char* name, int value) {
    static const char* const names[kVarCount] = {"march", "planes_fwd", "planes_bwd", "hashgrid_fwd", "hashgrid_bwd", "hash4d_bwd", "slice_plan",
                                                 "render_tail", "march_skew", "mlp_bwd", "level_kinds"};
    if (!name || value < 0) return NVSF_ERR_INVALID_ARG;
    for (int k = 0; k < kVarCount; ++k)
        if (strcmp(name, names[k]) == 0) {
            const int old = g_variant[k];
            g_variant[k] = value;
            return old;  // the previous value (>= 0), so that a caller can restore it
        }
    return NVSF_ERR_INVALID_ARG;
}
