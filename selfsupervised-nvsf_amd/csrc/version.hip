// Library identification (include/nvsf_hip.h: nvsf_version) and the test-only variant table (nvsf_test_variant).
#include "common.h"
#include <string.h>
NVSF_API const char* nvsf_version(void) { return "nvsf_hip 0.1.0 gfx950"; }

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
                                                 "render_tail", "march_skew", "mlp_bwd"};
    if (!name || value < 0) return NVSF_ERR_INVALID_ARG;
    for (int k = 0; k < kVarCount; ++k)
        if (strcmp(name, names[k]) == 0) {
            const int old = g_variant[k];
            g_variant[k] = value;
            return old;  // the previous value (>= 0), so that a caller can restore it
        }
    return NVSF_ERR_INVALID_ARG;
}
