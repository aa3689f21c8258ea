// Library identification (include/nvsf_hip.h: nvsf_version).
#include "common.h"
NVSF_API const char* nvsf_version(void) { return "nvsf_hip 0.1.0 gfx950"; }
